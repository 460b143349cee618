#!/usr/bin/env python3
"""Small-image latency through the Python API (BASELINE cfg 1 and friends): wall time per call of
denoise / AtrousTransform+sum / wow on numpy arrays, 512^2 .. 2048^2.

    python tools/bench_small.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavelets_amd as W

def t(fn, n=50):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3

for side, dt in ((512, np.float32), (1024, np.float32), (2048, np.float32), (512, np.float64), (2048, np.float64)):
    a = np.random.default_rng(0).standard_normal((side, side)).astype(dt)
    tr = W.AtrousTransform(W.B3spline)
    print(f"{side}^2 {np.dtype(dt).name}: denoise([5,3]) {t(lambda: W.denoise(a, [5, 3])):.3f} ms   "
          f"transform L=4 + np.sum {t(lambda: np.sum(tr(a, 4), axis=0)):.3f} ms   "
          f"wow {t(lambda: W.wow(a), 20):.3f} ms   "
          f"wow(bilateral=1) {t(lambda: W.wow(a, bilateral=1), 20):.3f} ms")
