#!/usr/bin/env python3
"""What the FIRST call of a process costs at 8192^2 (the one-shot `denoise(img)` user) and where it goes.

    python tools/first_call.py [side]

Prints the wall time of: loading the library, creating the context, the first and the second `denoise(img,
[5, 3])` numpy to numpy, and - in a second context-free pass - the pieces of the first call timed one by one
(plan creation, first upload, first kernels, first download)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
img = np.random.default_rng(0).standard_normal((side, side), dtype=np.float32)
t = time.perf_counter()


def lap(what, t0):
    t1 = time.perf_counter()
    print(f"{what:46s} {(t1 - t0) * 1e3:9.2f} ms", flush=True)
    return t1


import wavelets_amd as W
from wavelets_amd import _lib as L
t = lap("import wavelets_amd", t)
L.load()
t = lap("load libwatroo_hip.so", t)
ctx = L.default_context()
t = lap("context (HIP runtime init, stream, scratch)", t)
if "nosync" not in sys.argv:
    ctx.sync()
    t = lap("sync (joins the context's warm-up thread)", t)
if "warm_dma" in sys.argv or "warm_all" in sys.argv:
    # experiment: what a small warm-up moves out of the first call (DMA queues / staging of the runtime, code objects)
    small = np.zeros((256, 256), np.float32)
    wp = L.Plan(ctx, 256, 256, L.B3SPLINE, 2)
    wp.upload(L.PLANE_INPUT, small)
    if "warm_all" in sys.argv:
        W.denoise(small, [5, 3])
    wp.download(L.PLANE_INPUT)
    wp.close()
    t = lap("warm-up on a 256^2 image", t)
if "pieces" in sys.argv:
    plan = L.Plan(ctx, side, side, L.B3SPLINE, 2)
    t = lap("plan create", t)
    plan.upload(L.PLANE_INPUT, img)
    t = lap("first upload (pin + H2D 256 MiB)", t)
    plan.upload(L.PLANE_INPUT, img)
    t = lap("second upload", t)
    plan.decompose_sum(L.PLANE_INPUT, 2, L.PLANE_OUT)
    ctx.sync()
    t = lap("first decompose_sum (plane allocation, code load)", t)
    plan.decompose_sum(L.PLANE_INPUT, 2, L.PLANE_OUT)
    ctx.sync()
    t = lap("second decompose_sum", t)
    m = plan.abs_median(0)
    t = lap("first abs_median", t)
    m = plan.abs_median(0)
    t = lap("second abs_median", t)
    out = plan.download(L.PLANE_OUT)
    t = lap("first download (pinned result + D2H 256 MiB)", t)
    out = plan.download(L.PLANE_OUT)
    t = lap("second download", t)
    plan.close()
    t = lap("plan close", t)
if "trace" in sys.argv:
    # every C-ABI call of the first denoise, timed (a proxy in place of the ctypes library object)
    real = L.load()

    class Proxy:
        def __getattr__(self, name):
            fn = getattr(real, name)

            def timed(*a):
                t0 = time.perf_counter()
                r = fn(*a)
                dt = (time.perf_counter() - t0) * 1e3
                if dt > 0.2:
                    print(f"      {name:32s} {dt:9.2f} ms", flush=True)
                return r
            return timed
    L._lib = Proxy()
for i in range(4):
    if i == 2 and "trace" in sys.argv:
        L._lib = real
    out = W.denoise(img, [5, 3])
    t = lap(f"denoise(img, [5, 3]) call {i + 1}", t)
    del out
