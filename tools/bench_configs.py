#!/usr/bin/env python3
"""Secondary BASELINE.json configs (2, 3, 5): wall time of the public API call (host numpy in /
out, PCIe included) and the device time per kernel (HIP events).  Not the headline bench."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wavelets_amd as W            # noqa: E402
from wavelets_amd import _lib       # noqa: E402


def profile(label, fn, npix, reps=3):
    ctx = _lib.default_context()
    fn()                                            # warm-up (allocations, first launches)
    ctx.profile(True)
    ctx.profile_reset()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    wall = (time.perf_counter() - t) / reps
    ent = ctx.profile_entries()
    ctx.profile(False)
    dev = sum(ms for _, ms in ent.values()) / reps
    print(f"== {label}: wall {wall * 1e3:.1f} ms ({npix / wall / 1e6:.0f} Mpix/s incl. PCIe), "
          f"device {dev:.3f} ms ({npix / dev / 1e3:.0f} Mpix/s)")
    for k, (calls, ms) in sorted(ent.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:34s} calls/run {calls // reps:3d}  {ms / reps:9.3f} ms/run")


def main():
    which = sys.argv[1:] or ["2", "3", "5"]
    rng = np.random.default_rng(0)
    if "1" in which:
        a = rng.standard_normal((512, 512), dtype=np.float32)
        profile("cfg1 512^2 B3 L=4 denoise([5,3]) (README example)", lambda: W.denoise(a, [5, 3, 0, 0]), a.size, reps=20)
        profile("cfg1' 512^2 denoise(a, [5,3]) (2 scales)", lambda: W.denoise(a, [5, 3]), a.size, reps=20)
    if "2" in which:
        a = rng.standard_normal((4096, 4096), dtype=np.float32)
        profile("cfg2 4096^2 B3 L=6 decompose+sum", lambda: W.AtrousTransform(W.B3spline)(a, 6).sum(axis=0), a.size)
    if "3" in which:
        a = rng.standard_normal((8192, 8192), dtype=np.float32)
        def cfg3_api():
            c = W.AtrousTransform(W.Triangle)(a, 8)
            c.denoise([5, 3, 2])
            return np.sum(c, axis=0)

        def cfg3_fused():
            c = W.AtrousTransform(W.Triangle)(a, 8)
            return c._denoise_sum([5, 3, 2], write_back=True).download(-2)
        profile("cfg3 8192^2 Triangle L=8 + denoise([5,3,2]) + sum (reference call sequence)", cfg3_api, a.size)
        profile("cfg3 same, denoise fused into the plane sum (wt_denoise_sum)", cfg3_fused, a.size)
        profile("denoise(a,[5,3,2],Triangle) convenience (3 scales)", lambda: W.denoise(a, [5, 3, 2], W.Triangle), a.size)
    if "5" in which:
        side = int(os.environ.get("CFG5_SIDE", "8192"))
        a = (rng.standard_normal((side, side), dtype=np.float32)
             + 3 * np.sin(np.arange(side, dtype=np.float32) / 50.)[None, :])
        profile(f"cfg5 wow(bilateral=1, denoise_coefficients=[5,2]) {side}^2",
                lambda: W.wow(a, bilateral=1, denoise_coefficients=[5, 2]), a.size, reps=1)
    if "w" in which:
        side = int(os.environ.get("WOW_SIDE", "8192"))
        a = rng.standard_normal((side, side), dtype=np.float32)
        profile(f"wow(default) {side}^2", lambda: W.wow(a), a.size, reps=3)
        profile(f"wow(denoise_coefficients=[5,2]) {side}^2",
                lambda: W.wow(a, denoise_coefficients=[5, 2]), a.size, reps=3)


if __name__ == "__main__":
    main()
