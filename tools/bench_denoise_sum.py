#!/usr/bin/env python3
"""Device time of the fused threshold + plane-sum kernels (wt_denoise_sum / wt64_denoise_sum) at 8192^2:
four planes, three of them soft-thresholded, written back (the threshold step of cfg3).

    python tools/bench_denoise_sum.py [side]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wavelets_amd as W            # noqa: E402
from wavelets_amd import _lib as L  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = L.default_context()
rng = np.random.default_rng(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


for name, plan, dt in (("float32", L.Plan(ctx, side, side, L.TRIANGLE, 4), np.float32),
                       ("float64", L.Plan64(ctx, side, side, tuple(float(t) for t in W.Triangle.coefficients_1d), 4), np.float64)):
    for k in range(4):
        plan.upload(k, rng.standard_normal((side, side)).astype(dt))
    taus, wgts = [1.5, 0.9, 0.6], [1.0, 1.0, 1.0]
    if dt == np.float32:
        t = timed(lambda: plan.denoise_sum(4, taus, wgts, True, L.PLANE_NONE, True))
    else:
        t = timed(lambda: plan.denoise_sum(4, taus, wgts, True, L.PLANE_NONE, True))
    bpp = 8 * dt().itemsize                       # read 4 planes, write 3 + the sum
    print(f"{name}: {t:.4f} ms  ({side * side * bpp / t / 1e9:.2f} TB/s on the planes it touches)")
    plan.close()
