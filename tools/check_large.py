#!/usr/bin/env python3
"""Size-independent checks at the largest benchmark geometry, entirely on the device:
32768 x 32768 (4 GiB planes): sum of planes == input, DC preservation, and 8 virtual row strips
== unsharded (bitwise, compared on the device).  python tools/check_large.py [side]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavelets_amd import _lib as L
from wavelets_amd.parallel import partition_rows

side = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = L.default_context()
LEVEL = 6
rng = np.random.default_rng(0)
block = rng.standard_normal((2048, side), dtype=np.float32)
whole = L.Plan(ctx, side, side, L.B3SPLINE, LEVEL)
# upload in 2048-row blocks through a strip-shaped helper plan (no 4 GiB host array)
img_rows = []
helper = L.Plan(ctx, 2048, side, L.B3SPLINE, 0)
for i in range(side // 2048):
    helper.upload(L.PLANE_INPUT, block * np.float32(1 + 0.01 * i))
    helper.paste_into(whole, L.PLANE_INPUT, L.PLANE_INPUT, i * 2048, 0)
helper.close()
whole.decompose_sum(L.PLANE_INPUT, LEVEL, L.PLANE_OUT)      # what bench.py runs
D = L.PLANE_SCRATCH(6)
whole.binary("sub", L.PLANE_OUT, L.PLANE_INPUT, D)
s, s2, lo, hi = whole.reduce(D)
amax = max(abs(v) for v in whole.reduce(L.PLANE_INPUT)[2:])
print(f"reconstruction - input: min {lo:.3e} max {hi:.3e} (|input| max {amax:.3f})")
assert max(abs(lo), abs(hi)) <= 1e-5 * amax
tot_in = whole.reduce(L.PLANE_INPUT)[0]
tot_c = whole.reduce(LEVEL)[0]
print(f"mean(input) {tot_in / side / side:.6e}  mean(smooth) {tot_c / side / side:.6e}")
assert abs(tot_in - tot_c) / side / side < 1e-5

# 8 virtual strips vs unsharded, compared on the device plane by plane
k = 8
plans = []
for r, (row0, n) in enumerate(partition_rows(side, k)):
    p = L.Plan(ctx, side, side, L.B3SPLINE, LEVEL, row0=row0, nrows=n, rank=r, nranks=k)
    p.crop_from(whole, L.PLANE_INPUT, L.PLANE_INPUT, row0, 0)
    plans.append(p)
cur = L.PLANE_INPUT
FUSED = not os.environ.get('UNFUSED')
if not FUSED:
    whole.decompose(L.PLANE_INPUT, LEVEL, 0)
    whole.plane_sum(0, LEVEL + 1)
for i, (s0, ns, halo) in enumerate(L.schedule(L.B3SPLINE, LEVEL, FUSED)):
    nxt = LEVEL if s0 + ns == LEVEL else L.PLANE_SCRATCH(i & 1)
    for up, lo_ in zip(plans[:-1], plans[1:]):
        L.Plan.halo_exchange_local(up, lo_, cur, halo)
    for p in plans:
        if FUSED:
            p.decompose_pass_sum(cur, nxt, s0, ns, L.FLAG_FUSED | L.FLAG_NO_EXCHANGE, L.PLANE_OUT,
                                 first=i == 0, last=s0 + ns == LEVEL)
        else:
            p.decompose_pass(cur, nxt, s0, ns, L.FLAG_NO_EXCHANGE)
    cur = nxt
if not FUSED:
    for p in plans:
        p.plane_sum(0, LEVEL + 1)
worst = 0.0
for s in list(range(LEVEL + 1)) + [L.PLANE_OUT]:
    for p in plans:
        # bring the matching rows of the unsharded plane next to the strip's plane and subtract
        p.crop_from(whole, s, D, p.row0, 0)
        p.binary("sub", s, D, D)
        d = p.download(D)
        m = float(np.abs(d).max())
        if m > 0:
            bad = np.argwhere(d != 0)
            print(f"  plane {s} strip {p.rank}: max diff {m:.3e}, {len(bad)} px, rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()}")
        worst = max(worst, m)
print(f"8 virtual strips vs unsharded: max |difference| over all planes = {worst}")
assert worst == 0.0 or os.environ.get('NOFAIL')
print("check_large: OK")
