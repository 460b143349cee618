#!/usr/bin/env python3
"""Host-boundary timing: upload / decompose+sum / download of one 8192^2 float32 image, with
and without page-locking the caller's buffer (WT_PIN_THRESHOLD).  python tools/bench_pcie.py [side]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavelets_amd import _lib as L   # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = L.default_context()
img = np.random.default_rng(0).standard_normal((side, side), dtype=np.float32)
out = np.empty_like(img)
out[:] = 0          # touch the pages
plan = L.Plan(ctx, side, side, L.B3SPLINE, 6)
MB = img.nbytes / 1e6


def t(f, n=5):
    f()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def step():
    plan.decompose(L.PLANE_INPUT, 6, L.FLAG_FUSED)
    plan.plane_sum(0, 7, L.PLANE_OUT)
    ctx.sync()


up = t(lambda: plan.upload(L.PLANE_INPUT, img))
cm = t(step)
dn = t(lambda: plan.download(L.PLANE_OUT, out))
print(f"pin threshold {os.environ.get('WT_PIN_THRESHOLD', 'default')}: upload {up:.2f} ms "
      f"({MB / up:.1f} GB/s)  compute {cm:.2f} ms  download {dn:.2f} ms ({MB / dn:.1f} GB/s)  "
      f"end-to-end {side * side / (up + cm + dn) / 1e3:.0f} Mpix/s")

# one-shot behaviour: a fresh source array (touched by the producer) and a fresh np.empty target
for trial in range(3):
    src = img + np.float32(trial)            # new pages, written by the CPU
    t0 = time.perf_counter()
    plan.upload(L.PLANE_INPUT, src)
    t1 = time.perf_counter()
    dst = np.empty_like(img)                 # untouched pages
    t2 = time.perf_counter()
    plan.download(L.PLANE_OUT, dst)
    t3 = time.perf_counter()
    dst2 = np.empty_like(img)
    dst2[:] = 0                              # pre-faulted
    t4 = time.perf_counter()
    plan.download(L.PLANE_OUT, dst2)
    t5 = time.perf_counter()
    print(f"one-shot: upload fresh array {1e3 * (t1 - t0):.2f} ms; download into np.empty "
          f"{1e3 * (t3 - t2):.2f} ms; np.empty+memset {1e3 * (t4 - t3):.2f} ms; download into "
          f"pre-faulted {1e3 * (t5 - t4):.2f} ms")
    del src, dst, dst2

# what the drop-in API does: results come back in pooled page-locked arrays (_lib.host_empty)
import wavelets_amd as W   # noqa: E402
for trial in range(4):
    src = img + np.float32(trial)
    t0 = time.perf_counter()
    res = W.denoise(src, [5, 3])
    t1 = time.perf_counter()
    print(f"denoise(img, [5, 3]) numpy -> numpy, call {trial}: {1e3 * (t1 - t0):.2f} ms "
          f"({side * side / (t1 - t0) / 1e6:.0f} Mpix/s)")
    del res

# ... and with the noise level given: every threshold is known up front, the call is one pipelined pass
# (wt_denoise_sum_host: about one PCIe leg instead of two)
for trial in range(4):
    src = img + np.float32(trial)
    t0 = time.perf_counter()
    res = W.denoise(src, [5, 3, 0, 0, 0, 0], noise=1.0)
    t1 = time.perf_counter()
    print(f"denoise(img, [5, 3, 0, 0, 0, 0], noise=1.0) numpy -> numpy, call {trial}: {1e3 * (t1 - t0):.2f} ms "
          f"({side * side / (t1 - t0) / 1e6:.0f} Mpix/s)")
    del res
L.set_option("host_pipeline", 0)
for trial in range(2):
    src = img + np.float32(trial)
    t0 = time.perf_counter()
    res = W.denoise(src, [5, 3, 0, 0, 0, 0], noise=1.0)
    t1 = time.perf_counter()
    print(f"  the same, serial legs (host_pipeline off), call {trial}: {1e3 * (t1 - t0):.2f} ms")
    del res
L.set_option("host_pipeline", 1)
