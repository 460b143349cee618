import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavelets_amd as W
from oracle import atrous_numpy as O
np.set_printoptions(linewidth=200, precision=5, suppress=True)
shape, L = (37, 53), 2
a = np.random.default_rng(1).standard_normal(shape).astype(np.float32)
c = W.AtrousTransform(W.B3spline)(a, L).data
ref = O.atrous_standard(a, L, "b3spline")
bad = np.argwhere(np.abs(c - ref) > 1e-4)
print("bad", len(bad))
c1 = a - ref[0]
# vertical-only filtered image (symmetric border), taps 1/16,1/4,3/8,1/4,1/16
k = np.array([1, 4, 6, 4, 1], np.float32) / 16
pad = np.pad(a, ((2, 2), (0, 0)), mode="symmetric")
v0 = sum(k[j] * pad[j:j + shape[0]] for j in range(5))
for (pl, r, x) in bad[:12]:
    cen = c[pl, r, x] + c1[r, x]
    hits = np.argwhere(np.abs(a - cen) < 2e-6)
    hv = np.argwhere(np.abs(v0 - cen) < 2e-6)
    hc = np.argwhere(np.abs(c1 - cen) < 2e-6)
    print((pl, r, x), "got", c[pl, r, x], "ref", ref[pl, r, x], "implied cen", cen, "true cen", a[r, x],
          "img hits", hits.tolist()[:4], "v0 hits", hv.tolist()[:4], "c1 hits", hc.tolist()[:4])
