#!/usr/bin/env python3
"""float64 engine: device time of the standard transform + MAD + denoise + sum, and a large-size
check against the numpy oracle in float64.  python tools/bench_f64.py [side]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavelets_amd as W
from wavelets_amd import _lib as L
from oracle import atrous_numpy as O

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = L.default_context()
a = np.random.default_rng(0).standard_normal((side, side)) + 1e4
taps = tuple(W.B3spline.coefficients_1d)
plan = L.acquire_plan64(ctx, side, side, taps, 6)
plan.upload(L.PLANE_INPUT, a)
def generic():
    L.set_option("fused64", 0)
    plan.decompose(L.PLANE_INPUT, 6)
    L.set_option("fused64", 1)


for name, fn in (("decompose L=6", lambda: plan.decompose(L.PLANE_INPUT, 6)),
                 ("decompose+sum L=6", lambda: plan.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT)),
                 ("generic decomp.", generic),
                 ("abs_median", lambda: plan.abs_median(0)),
                 ("denoise 1 plane", lambda: plan.denoise(0, 1.0, 1.0, True)),
                 ("plane_sum 7", lambda: plan.plane_sum(0, 7))):
    fn(); ctx.sync()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t) / 10
    print(f"{side}^2 f64 {name:18s} {dt * 1e3:8.3f} ms   {side * side / dt / 1e9:7.2f} Gpix/s")
ctx.profile(True); ctx.profile_reset()
for _ in range(5):
    plan.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT)
for _ in range(5):
    plan.decompose(L.PLANE_INPUT, 6)
for k, (calls, ms) in ctx.profile_entries().items():
    print(f"    {k:24s} {calls:3d} launches  {ms / calls:7.4f} ms")
ctx.profile(False)
# correctness at size against the float64 oracle (1024 x 2048, L = 5: dilations up to 16)
b = np.random.default_rng(1).standard_normal((1024, 2048)) * 50 + 3e4
got = W.AtrousTransform(W.B3spline)(b, 5).data
ref = O.atrous_standard(b, 5, "b3spline")
print("1024x2048 L=5 max |gpu - oracle| / max|b| =", float(np.abs(got - ref).max() / np.abs(b).max()))
den = W.denoise(b, [5, 3, 2])
print("denoise max rel diff =", float(np.abs(den - O.denoise(b.copy(), [5, 3, 2], "b3spline")).max() / np.abs(b).max()))
