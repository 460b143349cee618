#!/usr/bin/env python3
"""Device time of the individual fused passes (HIP events), for schedule decisions.

    python tools/bench_passes.py [side] [family]

Times every pass the library builds - plain and with the carried sum - on a side x side image:
(s0, ns) = (0,3) (0,2) (3,3) (3,2) (3,1) (6,2) (6,1) and, for the 3-tap family, (0,4) (4,4)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavelets_amd import _lib as L  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fam_name = sys.argv[2] if len(sys.argv) > 2 else "b3spline"
fam = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam_name]
reps = 50 if side <= 4096 else 20
ctx = L.default_context()
plan = L.Plan(ctx, side, side, fam, 8)
plan.upload(L.PLANE_INPUT, np.random.default_rng(0).standard_normal((side, side), dtype=np.float32))
S0, S1 = L.PLANE_SCRATCH(0), L.PLANE_SCRATCH(1)
plan.copy(L.PLANE_INPUT, S0)
plan.fill(L.PLANE_OUT, 0.0)
passes = [(0, 3), (0, 2), (3, 3), (3, 2), (3, 1), (6, 2), (6, 1)]
if fam == L.TRIANGLE:
    passes += [(0, 4), (4, 4)]


def timeit(fn):
    for _ in range(5):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


# spin the clocks up
for _ in range(200):
    plan.decompose_pass(L.PLANE_INPUT, S1, 0, 2)
ctx.sync()
print(f"{side}x{side} {fam_name}: ms per launch (plain | carrying the sum | last pass of a sum)")
for s0, ns in passes:
    src = L.PLANE_INPUT if s0 == 0 else S0
    t_plain = timeit(lambda: plan.decompose_pass(src, S1, s0, ns))
    t_acc = timeit(lambda: plan.decompose_pass_sum(src, S1, s0, ns, L.FLAG_FUSED, L.PLANE_OUT, s0 == 0, False))
    t_sum = timeit(lambda: plan.decompose_pass_sum(src, S1, s0, ns, L.FLAG_FUSED, L.PLANE_OUT, s0 == 0, True))
    px = side * side
    print(f"  ({s0},{ns})  {t_plain:.4f}  {t_acc:.4f}  {t_sum:.4f}   "
          f"[{px * (4 * ns + 8) / t_plain / 1e6:.0f} | {px * (8 * ns + 8 + (0 if s0 == 0 else 4)) / t_acc / 1e6:.0f} GB/s of the pass's own traffic]")
plan.close()
