#!/usr/bin/env python3
"""Per-dilation device time of the float32 bilateral march (wt_bilateral2_kernel) at side x side: one bilateral
transform of this size's own number of scales, profiler entries split by dilation (WT_PROF_SCALES=1).
python tools/bench_bil.py [side] [steps]"""
import os, sys
os.environ["WT_PROF_SCALES"] = "1"
os.environ.setdefault("WT_NO_WOW_OVERLAP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wavelets_amd import _lib as L

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
level = int(np.round(np.log2(side) - np.log2(5)))
ctx = L.default_context()
img = (np.random.default_rng(0).standard_normal((side, side), dtype=np.float32)
       + 3 * np.sin(np.arange(side, dtype=np.float32) / 50.)[None, :]).astype(np.float32)
plan = L.Plan(ctx, side, side, L.B3SPLINE, level)
plan.upload(L.PLANE_INPUT, img)
plan.decompose_bilateral(L.PLANE_INPUT, level, [1.0] * level)
ctx.sync()
ctx.profile(True)
ctx.profile_reset()
for _ in range(steps):
    plan.decompose_bilateral(L.PLANE_INPUT, level, [1.0] * level)
ent = ctx.profile_entries()
ctx.profile(False)
tot = 0.0
for k, (calls, ms) in sorted(ent.items(), key=lambda kv: (len(kv[0]), kv[0])):
    if "bilateral" in k:
        print(f"{k:34s} {ms / calls:8.4f} ms")
        tot += ms / steps
print(f"bilateral kernels per transform: {tot:.3f} ms ({level} scales, {tot / level:.4f} ms per scale)")
plan.close()
