#!/usr/bin/env python3
"""cfg5 (wow(bilateral=1, denoise_coefficients=[5,2])) device-resident in float32 and float64, with the
per-kernel breakdown of the library's own profiler.  python tools/bench_wow64.py [side] [steps] [plain] [f32only]
plain: also wow() without bilateral filtering; f32only: stop after the float32 flow."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavelets_amd as W
from wavelets_amd import _lib as L
from wavelets_amd import utils as WU

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
plain = "plain" in sys.argv
ctx = L.default_context()
level = int(np.round(np.log2(side) - np.log2(5)))
img = (np.random.default_rng(0).standard_normal((side, side), dtype=np.float32)
       + 3 * np.sin(np.arange(side, dtype=np.float32) / 50.)[None, :]).astype(np.float32)


def run(f64, bilateral):
    sb = [1] * (level + 1) if bilateral else None
    tr = W.AtrousTransform(W.B3spline, bilateral=sb)
    if f64:
        plan = L.Plan64(ctx, side, side, tuple(float(t) for t in W.B3spline.coefficients_1d), level)
        plan.upload(L.PLANE_INPUT, img.astype(np.float64))
    else:
        plan = L.Plan(ctx, side, side, L.B3SPLINE, level)
        plan.upload(L.PLANE_INPUT, img)
    co = W.Coefficients(plan, W.B3spline(2), sb)

    def transform():
        if not f64 or not bilateral or hasattr(plan, "decompose_bilateral"):
            return tr._run(plan, level)
        cur = L.PLANE_INPUT                      # the per-scale sequence of AtrousTransform._call_f64
        for s in range(level):
            nxt = level if s == level - 1 else L.PLANE_SCRATCH(s & 1)
            plan.local_variance(cur, L.PLANE_SCRATCH(4), s, 1.0, 1.0)
            plan.bilateral_conv(cur, L.PLANE_SCRATCH(4), nxt, s, 0)
            plan.binary("sub", cur, nxt, s)
            cur = nxt

    def step():
        transform()
        co.noise = None
        WU._wow_device(co, level, [], True, [5, 2], True, False, 3.2, None, None, 0)
    step()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    ms = (time.perf_counter() - t) / steps * 1e3
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(2):
        step()
    ent = ctx.profile_entries()
    ctx.profile(False)
    print(f"{side}^2 L={level} {'float64' if f64 else 'float32'} bilateral={bilateral}: {ms:8.3f} ms/step", flush=True)
    tot = 0.0
    for k, (calls, kms) in sorted(ent.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:34s} {calls // 2:3d} launches/step  {kms / calls:8.4f} ms each  {kms / 2:8.3f} ms/step")
        tot += kms / 2
    print(f"    (profiled kernels: {tot:.3f} ms/step)")
    co._plan = None
    plan.close()
    return ms


m32 = run(False, True)
if "f32only" in sys.argv:
    sys.exit(0)
m64 = run(True, True)
print(f"float64 / float32 (bilateral) = {m64 / m32:.2f}")
if plain:
    p32 = run(False, False)
    p64 = run(True, False)
    print(f"float64 / float32 (plain) = {p64 / p32:.2f}")
