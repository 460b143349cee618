#!/usr/bin/env python3
"""Headline step (b3spline L=6 decompose + plane sum, device-resident) at widths that are not multiples of 4, per pixel
against 8192^2: the cost of the generic addressing of the fused passes.  python tools/bench_odd.py [HxW ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wavelets_amd import _lib as L

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8192, 8192), (8192, 8190), (8190, 8190), (8191, 8191), (8192, 8193),
                                                                          (3066, 3066), (3068, 3068), (4096, 4096), (4095, 4095)]
ctx = L.default_context()
base = None
for (h, w) in shapes:
    plan = L.Plan(ctx, h, w, L.B3SPLINE, 6)
    plan.upload(L.PLANE_INPUT, np.random.default_rng(0).standard_normal((h, w), dtype=np.float32))
    for _ in range(5):
        plan.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT, L.FLAG_FUSED)
    ctx.sync()
    ms = []
    for _ in range(20):
        ctx.timer_start()
        plan.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT, L.FLAG_FUSED)
        ms.append(ctx.timer_stop())
    plan.close()
    med = float(np.median(ms))
    per = med / (h * w)
    base = base or per
    print(f"{h}x{w}: {med:.4f} ms  {h * w / med / 1e3:9.0f} Mpix/s  per pixel x{per / base:.3f}", flush=True)
