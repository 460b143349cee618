"""What the riding histogram costs the first fused pass: wt_fused<d1xN> against wt_fused_hist<d1xN> (the pass that also
histograms |w_0| for the exact median, FLAG_MEDIAN_HIST) at 8192^2, both families, on N(0,1) data and on an image
with structure.  python tools/bench_hist_pass.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavelets_amd import _lib as L  # noqa: E402

ctx = L.default_context()
side = 8192
rng = np.random.default_rng(0)
imgs = {"gauss": rng.standard_normal((side, side), dtype=np.float32)}
imgs["struct"] = (imgs["gauss"] + 3 * np.sin(np.arange(side, dtype=np.float32) / 50.0)[None, :]).astype(np.float32)
for fam, level, name in ((L.B3SPLINE, 6, "b3 L=6"), (L.TRIANGLE, 8, "triangle L=8")):
    plan = L.Plan(ctx, side, side, fam, level)
    for tag, img in imgs.items():
        plan.upload(L.PLANE_INPUT, img)
        for flags, what in ((L.FLAG_FUSED, "plain"), (L.FLAG_FUSED | L.FLAG_MEDIAN_HIST, "hist")):
            for _ in range(3):
                plan.decompose(L.PLANE_INPUT, level, flags)
            ctx.profile(True)
            ctx.profile_reset()
            for _ in range(10):
                plan.decompose(L.PLANE_INPUT, level, flags)
            ent = ctx.profile_entries()
            ctx.profile(False)
            first = [(k, v) for k, v in ent.items() if "d1x" in k]
            print(f"{name} {tag} {what}: " + ", ".join(f"{k} {v[1] / max(v[0], 1):.4f} ms" for k, v in first), flush=True)
    plan.close()
