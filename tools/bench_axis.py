"""Device time of the tiled axis filters (wt_axis.h) by axis: 17 taps at 2048^2 (the round-4 review's case) and
at 8192^2, float32 and float64.  python tools/bench_axis.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavelets_amd import _lib as L  # noqa: E402


def main():
    ctx = L.default_context()
    rng = np.random.default_rng(1)
    taps = np.hanning(19)[1:-1]
    taps = taps / taps.sum()
    b3 = (1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16)
    for side in (2048, 8192):
        for f64 in (False, True):
            a = rng.standard_normal((side, side), dtype=np.float32)
            plan = L.Plan64(ctx, side, side, b3, 0) if f64 else L.Plan(ctx, side, side, L.B3SPLINE, 0)
            A, T1, B = L.PLANE_SCRATCH(2), L.PLANE_SCRATCH(3), L.PLANE_SCRATCH(4)
            plan.upload(A, a.astype(np.float64) if f64 else a)
            for d in (1, 16):
                o = np.arange(17) * d - (16 * d + 1) // 2
                res = []
                for axis, src, dst in ((2, A, T1), (1, T1, B)):
                    for _ in range(3):
                        plan.axis_filter(src, dst, axis, o, taps)
                    ctx.sync()
                    ctx.timer_start()
                    for _ in range(50):
                        plan.axis_filter(src, dst, axis, o, taps)
                    res.append(ctx.timer_stop() / 50)
                gb = side * side * (8 if f64 else 4) * 2 / 1e9
                print(f"{side}^2 {'f64' if f64 else 'f32'} d={d}: x {res[0]*1e3:.1f} us ({gb/res[0]:.0f} GB/s... {gb/res[0]/1e-3/1e3:.2f} TB/s), "
                      f"y {res[1]*1e3:.1f} us ({gb/res[1]/1e-3/1e3:.2f} TB/s)", flush=True)
            plan.close()


if __name__ == "__main__":
    main()
