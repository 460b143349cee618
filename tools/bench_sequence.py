#!/usr/bin/env python3
"""A sequence of frames, numpy to numpy: denoise(frame, [5, 3]) in a loop against sequence.denoise_many on 1-4 lanes
(8192^2 float32 by default; PCIe both ways).  python tools/bench_sequence.py [side] [frames] [noise]
noise: give the noise level (the pipelined host call of denoise()) instead of the per-frame MAD estimate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavelets_amd as W

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
noise = 1.0 if "noise" in sys.argv else None
rng = np.random.default_rng(0)
base = rng.standard_normal((side, side), dtype=np.float32)
frames = [base + np.float32(i) for i in range(n)]          # pageable arrays, as a user holds them


def run(label, fn, reps=2):
    best = None
    for _ in range(reps + 1):                               # (first round: plans, lanes, pinned blocks)
        t = time.perf_counter()
        res = fn()
        dt = (time.perf_counter() - t) / n * 1e3
        best = dt if best is None or _ > 0 and dt < best else best
        del res
    print(f"{label:34s} {best:7.2f} ms per frame  {side * side / best / 1e3:9.0f} Mpix/s", flush=True)
    return best


loop = run("loop of denoise()", lambda: [W.denoise(f, [5, 3], noise=noise) for f in frames])
for lanes in (1, 2, 3, 4):
    run(f"denoise_many, {lanes} lane(s)", lambda: W.denoise_many(frames, [5, 3], noise=noise, lanes=lanes))
ref = W.denoise(frames[3], [5, 3], noise=noise)
got = W.denoise_many(frames, [5, 3], noise=noise)
print("bitwise equal to the per-call result:", all(np.array_equal(got[3], ref) for _ in (0,)))
