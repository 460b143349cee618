#!/usr/bin/env python3
"""sha1 of the planes + carried sum of decompose_sum for a few shapes: compare libraries built with
different options (WATROO_HIP_LIB=variants/x.so python tools/lib_digest.py) bit for bit."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wavelets_amd import _lib as L

ctx = L.default_context()
for (H, W), fam, level in (((8192, 8192), L.B3SPLINE, 6), ((4096, 4096), L.TRIANGLE, 8), ((700, 1100), L.B3SPLINE, 5),
                           ((333, 1001), L.TRIANGLE, 3), ((1500, 2900), L.B3SPLINE, 2)):
    a = np.random.default_rng(H + W).standard_normal((H, W), dtype=np.float32)
    plan = L.Plan(ctx, H, W, fam, level)
    plan.upload(L.PLANE_INPUT, a)
    plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, L.FLAG_FUSED)
    h = hashlib.sha1()
    for s in range(level + 1):
        h.update(plan.download(s).tobytes())
    h.update(plan.download(L.PLANE_OUT).tobytes())
    plan.decompose(L.PLANE_INPUT, level, L.FLAG_FUSED | 16)
    h2 = hashlib.sha1()
    for s in range(level + 1):
        h2.update(plan.download(s).tobytes())
    print(f"{H}x{W} fam{fam} L{level}: sum-pass {h.hexdigest()[:16]}  plain+hist {h2.hexdigest()[:16]}  median {plan.abs_median(0)!r}")
    plan.close()
