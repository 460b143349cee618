#!/usr/bin/env python3
"""decompose_sum vs decompose + plane_sum: bitwise check on assorted shapes, then timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavelets_amd import _lib as L

ctx = L.default_context()
rng = np.random.default_rng(0)
bad = 0
for fam in (L.B3SPLINE, L.TRIANGLE):
    for (H, W) in ((37, 53), (300, 1000), (1, 700), (513, 129), (1024, 4096), (2000, 3001)):
        for level in (1, 2, 3, 4, 5, 6, 7, 8):
            a = rng.standard_normal((H, W), dtype=np.float32)
            p = L.Plan(ctx, H, W, fam, level)
            p.upload(L.PLANE_INPUT, a)
            p.decompose(L.PLANE_INPUT, level, L.FLAG_FUSED)
            ref = [p.download(s) for s in range(level + 1)]
            p.plane_sum(0, level + 1, L.PLANE_OUT)
            rsum = p.download(L.PLANE_OUT)
            for s in range(level + 1):
                p.fill(s, np.nan)
            p.fill(L.PLANE_OUT, np.nan)
            p.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, L.FLAG_FUSED)
            ok = all(np.array_equal(p.download(s), ref[s]) for s in range(level + 1))
            oks = np.array_equal(p.download(L.PLANE_OUT), rsum)
            if not (ok and oks):
                bad += 1
                d = p.download(L.PLANE_OUT)
                print(f"MISMATCH fam {fam} {H}x{W} L={level}: planes {ok} sum {oks} "
                      f"max|d| {np.nanmax(np.abs(d - rsum)):.3e} nan {np.isnan(d).sum()}")
            p.close()
print("mismatches:", bad)
side = 8192
p = L.Plan(ctx, side, side, L.B3SPLINE, 6)
p.upload(L.PLANE_INPUT, rng.standard_normal((side, side), dtype=np.float32))
def two():
    p.decompose(L.PLANE_INPUT, 6, L.FLAG_FUSED); p.plane_sum(0, 7, L.PLANE_OUT)
def one():
    p.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT, L.FLAG_FUSED)
for name, f in (("two calls", two), ("decompose_sum", one), ("two calls", two), ("decompose_sum", one)):
    for _ in range(5): f()
    ctx.sync(); t = time.perf_counter()
    for _ in range(30): f()
    ctx.sync(); ms = (time.perf_counter() - t) / 30 * 1e3
    print(f"{name}: {ms:.4f} ms  {side * side / ms / 1e3:.0f} Mpix/s")
ctx.profile(True); ctx.profile_reset()
for _ in range(10): one()
for k, (c, ms) in ctx.profile_entries().items(): print(f"   {k}: {ms / c:.4f} ms")
