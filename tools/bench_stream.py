#!/usr/bin/env python3
"""Device time (HIP events) of the streaming kernels - reductions, select passes, thresholds - at
side x side, with the bytes each launch has to move:  python tools/bench_stream.py [side]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavelets_amd import _lib as L  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = L.default_context()
img = np.random.default_rng(0).standard_normal((side, side), dtype=np.float32)
px = side * side


def prof(label, fn, bytes_per_px, reps=10):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps):
        fn()
    ent = ctx.profile_entries()
    ctx.profile(False)
    tot = sum(ms for _, ms in ent.values()) / reps
    print(f"{label:44s} {tot:8.4f} ms  {px * bytes_per_px / tot / 1e6:7.0f} GB/s   " +
          "  ".join(f"{k}={ms / reps:.4f}" for k, (c, ms) in ent.items()))


p = L.Plan(ctx, side, side, L.B3SPLINE, 3)
p.upload(L.PLANE_INPUT, img)
p.decompose(L.PLANE_INPUT, 3, L.FLAG_FUSED)
prof("f32 reduce (4 B/px)", lambda: p.reduce(0), 4)
prof("f32 abs_median, 3 passes (12 B/px)", lambda: p.abs_median(0), 12)
prof("f32 plane_sum 4 planes (20 B/px)", lambda: p.plane_sum(0, 4, L.PLANE_OUT), 20)
prof("f32 denoise 1 plane (8 B/px)", lambda: p.denoise(1, 0.7, 1.0, True, L.PLANE_NONE), 8)
p.close()

q = L.Plan64(ctx, side, side, (1 / 4, 1 / 2, 1 / 4), 8)
q.upload(L.PLANE_INPUT, img.astype(np.float64))
q.decompose(L.PLANE_INPUT, 8)
prof("f64 reduce (8 B/px)", lambda: q.reduce(0), 8)
prof("f64 abs_median: hist + collect (16 B/px)", lambda: q.abs_median(0), 16)
L.set_option("select64_list", 0)
prof("f64 abs_median: 6 radix passes (48+ B/px)", lambda: q.abs_median(0), 48)
L.set_option("select64_list", 1)
prof("f64 denoise 1 plane (16 B/px)", lambda: q.denoise(1, 0.7, 1.0, True, L.PLANE_NONE), 16)
prof("f64 hard threshold 1 plane (16 B/px)", lambda: q.denoise(1, 0.7, 1.0, False, L.PLANE_NONE), 16)
prof("f64 plane_sum 9 planes (80 B/px)", lambda: q.plane_sum(0, 9, L.PLANE_OUT), 80)
prof("f64 denoise_sum 4 planes, 3 thresholded, written back (64 B/px)",
     lambda: q.denoise_sum(4, [0.7, 0.5, 0.3], [1, 1, 1], True, L.PLANE_NONE, True), 64)
prof("f64 denoise_sum 9 planes, 3 thresholded, not written back (80 B/px)",
     lambda: q.denoise_sum(9, [0.7, 0.5, 0.3], [1, 1, 1], True, L.PLANE_NONE, False), 80)
q.close()
