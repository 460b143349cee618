#!/usr/bin/env python3
"""Cost of the overlapped schedule's launch split on ONE GPU (no exchange): a middle strip of a
tall image runs decompose_sum whole and split into edge rows + interior rows (option split_dry).

    python tools/bench_split.py [nrows=4096] [W=32768]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wavelets_amd import _lib as L

nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
ctx = L.default_context()
plan = L.Plan(ctx, 3 * nrows, W, L.B3SPLINE, 6, row0=nrows, nrows=nrows, rank=1, nranks=3)
rng = np.random.default_rng(0)
plan.upload(L.PLANE_INPUT, rng.standard_normal((nrows, W), dtype=np.float32))
flags = L.FLAG_FUSED | L.FLAG_NO_EXCHANGE
ref = None
for reserve in (None, 0, 16, 32):
    L.set_option("split_dry", 0 if reserve is None else 1)
    if reserve is not None:
        L.set_option("overlap_reserve", reserve)
    for _ in range(20):
        plan.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT, flags)
    ctx.sync()
    ctx.profile(True); ctx.profile_reset()
    t = time.perf_counter()
    for _ in range(20):
        plan.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT, flags)
    ctx.sync()
    dt = (time.perf_counter() - t) / 20
    prof = ctx.profile_entries(); ctx.profile(False)
    out = plan.download(L.PLANE_OUT)
    if ref is None:
        ref = out
    same = bool(np.array_equal(out, ref))
    print(f"{'whole' if reserve is None else 'split reserve=%d' % reserve}: {dt * 1e3:.3f} ms/step "
          f"({nrows * W / dt / 1e9:.1f} Gpix/s)  identical={same}  "
          + "  ".join(f"{k}: {c // 20}x{ms / c:.3f}" for k, (c, ms) in prof.items()))
