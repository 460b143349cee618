#!/usr/bin/env python3
"""Device time of the exact-median select passes (wt_hist_kernel) on a Gaussian plane."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wavelets_amd import _lib as L
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = L.default_context()
plan = L.Plan(ctx, side, side, L.B3SPLINE, 1)
plan.upload(0, np.random.default_rng(0).standard_normal((side, side), dtype=np.float32) * 0.3)
for _ in range(5):
    plan.abs_median(0)
ctx.profile(True); ctx.profile_reset()
for _ in range(20):
    plan.abs_median(0)
for k, (calls, ms) in ctx.profile_entries().items():
    print(f"{k:24s} {calls:3d} launches {ms / calls:7.4f} ms  {side * side * 4 / (ms / calls) / 1e9:7.1f} TB/s-equivalent" if "hist" in k else f"{k:24s} {calls:3d} launches {ms / calls:7.4f} ms")
