#!/usr/bin/env python3
"""Per-scale device time of the per-scale operators (chain-march kernels) at 8192^2."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavelets_amd import _lib
L = _lib
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = L.default_context()
plan = L.Plan(ctx, side, side, L.B3SPLINE, 1)
plan.upload(L.PLANE_INPUT, np.random.default_rng(0).standard_normal((side, side), dtype=np.float32))
S3, S4 = L.PLANE_SCRATCH(3), L.PLANE_SCRATCH(4)
plan.fill(S4, 1.0)
plan.copy(L.PLANE_INPUT, 0)
ops = {
    "smooth": lambda s: plan.smooth(L.PLANE_INPUT, S3, s),
    "smooth_sq": lambda s: plan.smooth(L.PLANE_INPUT, S3, s, True),
    "decomp": lambda s: plan.atrous_scale(L.PLANE_INPUT, S3, 0, s),
    "variance": lambda s: plan.local_variance(L.PLANE_INPUT, S3, s),
    "wow": lambda s: plan.wow_scale(0, s, 0.0, True, L.PLANE_NONE, 1.0, L.PLANE_NONE),
    "wow_tau": lambda s: plan.wow_scale(0, s, 1.0, True, L.PLANE_NONE, 1.0, L.PLANE_NONE),
    "bilateral": lambda s: plan.bilateral_conv(L.PLANE_INPUT, S4, S3, s),
}
print("scale " + " ".join(f"{k:>10s}" for k in ops))
for s in range(0, 11):
    row = []
    for name, fn in ops.items():
        fn(s); ctx.sync()
        ctx.timer_start()
        for _ in range(5):
            fn(s)
        row.append(ctx.timer_stop() / 5)
    print(f"{s:5d} " + " ".join(f"{v:10.3f}" for v in row))
