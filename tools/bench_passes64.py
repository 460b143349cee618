#!/usr/bin/env python3
"""Device time of the float64 fused passes (HIP events), for schedule / workgroup-shape decisions.

    python tools/bench_passes64.py [side] [family]

Times the passes of the float64 engine - plain, carrying the sum, last pass of a sum - on a
side x side image (the float64 twin of tools/bench_passes.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wavelets_amd as W            # noqa: E402
from wavelets_amd import _lib as L  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fam_name = sys.argv[2] if len(sys.argv) > 2 else "triangle"
cls = {"b3spline": W.B3spline, "triangle": W.Triangle}[fam_name]
reps = 20
ctx = L.default_context()
plan = L.Plan64(ctx, side, side, tuple(float(t) for t in cls.coefficients_1d), 8)
plan.upload(L.PLANE_INPUT, np.random.default_rng(0).standard_normal((side, side)))
S0, S1 = L.PLANE_SCRATCH(0), L.PLANE_SCRATCH(1)
plan.copy(L.PLANE_INPUT, S0)
plan.fill(L.PLANE_OUT, 0.0)
passes = [(0, 3), (0, 2), (3, 3), (3, 2), (6, 2)]
if fam_name == "triangle":
    passes += [(0, 4), (4, 4)]


def timeit(fn):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


for _ in range(100):                       # spin the clocks up
    plan.decompose_pass(L.PLANE_INPUT, S1, 0, 2)
ctx.sync()
print(f"{side}x{side} float64 {fam_name}: ms per launch (plain | carrying the sum | last pass of a sum)")
for s0, ns in passes:
    src = L.PLANE_INPUT if s0 == 0 else S0
    t_plain = timeit(lambda: plan.decompose_pass(src, S1, s0, ns))
    t_acc = timeit(lambda: plan.decompose_pass_sum(src, S1, s0, ns, L.FLAG_FUSED, L.PLANE_OUT, s0 == 0, False))
    t_sum = timeit(lambda: plan.decompose_pass_sum(src, S1, s0, ns, L.FLAG_FUSED, L.PLANE_OUT, s0 == 0, True))
    print(f"  ({s0},{ns})  {t_plain:.4f}  {t_acc:.4f}  {t_sum:.4f}")
plan.close()
