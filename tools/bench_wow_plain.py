#!/usr/bin/env python3
"""Device-resident utils.wow(img, denoise_coefficients=[5, 2]) (no bilateral) at 8192^2, this
size's own n_scales (11): per-kernel times.  python tools/bench_wow_plain.py [side]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavelets_amd as WA
from wavelets_amd import _lib as L, utils as WU

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
level = 11 if side >= 8192 else int(np.log2(side)) - 2
ctx = L.default_context()
rng = np.random.default_rng(0)
a = rng.standard_normal((side, side), dtype=np.float32) + 3 * np.sin(np.arange(side, dtype=np.float32) / 37.)[None, :]
plan = L.acquire_plan(ctx, side, side, L.B3SPLINE, level)
plan.upload(L.PLANE_INPUT, a)
T = WA.AtrousTransform(WA.B3spline)
c = WA.Coefficients(plan, WA.B3spline(2))

def step():
    T._run(plan, level)
    c.noise = None
    WU._wow_device(c, level, [], True, [5, 2], True, False, 3.2, None, None, 0)

for _ in range(20):
    step()
ctx.sync()
ctx.profile(True); ctx.profile_reset()
t = time.perf_counter()
n = 20
for _ in range(n):
    step()
ctx.sync()
dt = (time.perf_counter() - t) / n
prof = ctx.profile_entries(); ctx.profile(False)
print(f"wow plain {side}^2 L={level}: {dt * 1e3:.3f} ms/step  {side * side / dt / 1e9:.2f} Gpix/s")
for k, (cnt, ms) in prof.items():
    print(f"   {k:32s} {cnt // n:3d} x {ms / cnt:.4f} ms = {ms / n:.3f}")
