#!/usr/bin/env python3
"""Random differential test of the lattice kernel against the chain kernel (bitwise), all
single-scale modes, dilations 64..4096, random shapes with W % 4 == 0.  python tools/fuzz_lattice.py [n] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavelets_amd import _lib as L   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = L.default_context()
S3, S4, G = L.PLANE_SCRATCH(6), L.PLANE_SCRATCH(7), L.PLANE_SCRATCH(4)
fails = 0
for case in range(n):
    H = int(rng.integers(1, 1500))
    Wd = 4 * int(rng.integers(1, 1300))
    s = int(rng.integers(6, 13))
    fam = (L.B3SPLINE, L.TRIANGLE)[int(rng.integers(0, 2))]
    a = rng.standard_normal((H, Wd), dtype=np.float32)
    outs = []
    for lat in (1, 0):
        L.set_option("lattice_kernel", lat)
        p = L.Plan(ctx, H, Wd, fam, 1)
        p.upload(L.PLANE_INPUT, a)
        p.upload(0, a)
        p.upload(G, 0.25 * a)
        res = []
        p.smooth(L.PLANE_INPUT, S3, s); res.append(p.download(S3))
        p.smooth(L.PLANE_INPUT, S3, s, True); res.append(p.download(S3))
        p.local_variance(L.PLANE_INPUT, S3, s, 1.5, 2.0); res.append(p.download(S3))
        p.atrous_scale(L.PLANE_INPUT, S3, S4, s); res += [p.download(S3), p.download(S4)]
        p.wow_scale(0, s, 1.1, True, L.PLANE_NONE, 0.8, G); res += [p.download(0), p.download(G)]
        outs.append(res)
        p.close()
    L.set_option("lattice_kernel", 1)
    if not all(np.array_equal(x, y) for x, y in zip(*outs)):
        fails += 1
        print(f"FAIL case {case}: {H}x{Wd} s={s} fam={fam}")
print(f"fuzz_lattice: {n} cases, {fails} failures")
sys.exit(1 if fails else 0)
