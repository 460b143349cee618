#!/usr/bin/env python3
"""An integer image (int16, what a FITS frame holds) through denoise() and wow(): numpy in, numpy out,
steady state - widened on the device (wt64_upload_int) against promoted to float64 on the host first.

    python tools/bench_int_input.py [side]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wavelets_amd as W            # noqa: E402
from wavelets_amd import _lib as L  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = L.default_context()
rng = np.random.default_rng(0)
img = (1000 + 80 * rng.standard_normal((side, side))).astype(np.int16)


def timed(fn, n=5):
    for _ in range(3):
        r = fn(); del r
    ctx.sync()
    t = time.perf_counter()
    for _ in range(n):
        r = fn(); del r
    ctx.sync()
    return (time.perf_counter() - t) / n * 1e3


import wavelets_amd._lib as LIB     # noqa: E402


def host_promotion(on):
    """switch the device widening off (the host astype of earlier rounds) / on"""
    L.Plan64._INT_CODES = {} if on else LIB._ELEM_CODES
    LIB.device_widens = (lambda dt: False) if on else keep_widens
    W.wavelets._lib.device_widens = LIB.device_widens


keep_widens = LIB.device_widens
images = (("int16 (float64 engine)", img), ("uint8 (float32 engine)", np.clip(img // 8, 0, 255).astype(np.uint8)),
          ("'>i2' raw FITS (float32 engine)", img.astype(">i2")))
for label, im in images:
    for name, fn in (("denoise(img, [5, 3, 2], Triangle)", lambda: W.denoise(im, [5, 3, 2], W.Triangle)), ("wow(img)", lambda: W.wow(im))):
        host_promotion(False)
        t_dev = timed(fn)
        host_promotion(True)
        t_host = timed(fn)
        host_promotion(False)
        print(f"{side}^2 {label}  {name}: {t_dev:.2f} ms widened on the device, {t_host:.2f} ms converted on the host")
