#!/usr/bin/env python3
"""PCIe-inclusive time of the drop-in transform + synthesis call (numpy in, numpy out) at 8192^2:
the serial legs against wt_decompose_sum_host for several block sizes, with a pageable and with a
page-locked input array.  python tools/bench_pcie_pipe.py [side]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wavelets_amd import _lib as L

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = L.default_context()
img = np.random.default_rng(0).standard_normal((side, side), dtype=np.float32)
plan = L.Plan(ctx, side, side, L.B3SPLINE, 6)
pinned = L.host_empty((side, side), ctx)
pinned[...] = img
out = L.host_empty((side, side), ctx)


def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3


def serial(src):
    plan.upload(L.PLANE_INPUT, src)
    plan.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT)
    plan.download(L.PLANE_OUT, out)


for name, src in (("pageable", img), ("page-locked", pinned)):
    lo, med = timed(lambda: serial(src))
    print(f"{side}^2 {name:12s} serial legs                 min {lo:7.3f} ms  median {med:7.3f} ms  {side * side / lo / 1e3:8.0f} Mpix/s")
    for block in (0, 2048, 1024, 512, 256, 128):
        lo, med = timed(lambda: plan.decompose_sum_host(src, 6, L.PLANE_OUT, out=out, block_rows=block))
        print(f"{side}^2 {name:12s} pipelined, block_rows {block:5d}  min {lo:7.3f} ms  median {med:7.3f} ms  {side * side / lo / 1e3:8.0f} Mpix/s")
t = time.perf_counter(); plan.upload(L.PLANE_INPUT, pinned); up = time.perf_counter() - t
t = time.perf_counter(); plan.download(L.PLANE_OUT, out); dn = time.perf_counter() - t
print(f"one leg: upload {up * 1e3:.3f} ms ({img.nbytes / up / 1e9:.1f} GB/s), download {dn * 1e3:.3f} ms ({img.nbytes / dn / 1e9:.1f} GB/s)")
# full duplex? an upload on one context beside a download on another (two host threads; ctypes
# releases the GIL and every context has its own stream and lock)
import threading
ctx2 = L.Context(0)
plan2 = L.Plan(ctx2, side, side, L.B3SPLINE, 0)
plan2.upload(L.PLANE_INPUT, pinned)
out2 = L.host_empty((side, side), ctx2)
plan2.download(L.PLANE_INPUT, out2)
N = 4
def ups():
    for _ in range(N):
        plan.upload(L.PLANE_INPUT, pinned)
def downs():
    for _ in range(N):
        plan2.download(L.PLANE_INPUT, out2)
t = time.perf_counter(); ups(); t_up = time.perf_counter() - t
t = time.perf_counter(); downs(); t_dn = time.perf_counter() - t
a, b = threading.Thread(target=ups), threading.Thread(target=downs)
t = time.perf_counter(); a.start(); b.start(); a.join(); b.join(); t_both = time.perf_counter() - t
print(f"{N} uploads alone {t_up * 1e3:.2f} ms, {N} downloads alone {t_dn * 1e3:.2f} ms, both at once {t_both * 1e3:.2f} ms "
      f"({2 * N * img.nbytes / t_both / 1e9:.1f} GB/s aggregate)")
