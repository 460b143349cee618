#!/usr/bin/env python3
"""Randomised differential test of the round-6 paths on the GPU box (run by tests/test_gpu_fuzz.py).

    python tools/fuzz_round6.py [n_cases] [seed] [only_case]

Per case one of:
  0  bilateral march, 8-byte operand pairs against the generic two-load path (bitwise): random shapes incl. odd widths,
     images smaller than the dilated kernel (several bounces), both families, sigma lists, bilateral_scaling - and the
     transform against the numpy oracle at the bilateral tolerance
  1  fused passes, fast addressing against the generic one at ANY width (bitwise), float32 and float64, planes + carried sum
  2  sequences: denoise_many / wow_many on 2-4 lanes against the per-call loop (bitwise), random dtypes (float32,
     float64, int16), noise given or estimated, frames fed from a generator
  3  exact median behind the riding histogram with register counters for the out-of-window keys: distributions that put
     nearly everything below / above / inside the window (constant images, heavy tails, two-valued images, ties) against
     np.median(|w_0|) of the downloaded plane (exact), float32 and float64
One line per failing case (replayable: same count and seed, or the case number as third argument) + a summary."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import atrous_numpy as O        # noqa: E402
import wavelets_amd as W                    # noqa: E402
from wavelets_amd import _lib as L          # noqa: E402


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def case_bilateral_pairs(rng):
    fam = (W.B3spline, W.Triangle)[int(rng.integers(2))]
    kind = int(rng.integers(4))
    if kind == 0:
        H, Wd = int(rng.integers(3, 40)), int(rng.integers(1, 40))           # smaller than the kernel: several bounces
    elif kind == 1:
        H, Wd = int(rng.integers(40, 300)), int(rng.integers(40, 700))
    else:
        H, Wd = int(rng.integers(200, 900)), int(rng.integers(129, 1500))
    level = int(rng.integers(1, 7))
    # (pedestal of the order of the noise: sdev_loc's conv(I^2) - conv(I)^2 cancels in float32, in the reference as here,
    #  and on a large pedestal the two roundings of that difference - not the kernels - set the distance to the oracle)
    scale = rng.uniform(0.5, 30)
    a = (rng.standard_normal((H, Wd)) * scale + scale * rng.uniform(-3, 3)).astype(np.float32)
    sig = [float(rng.uniform(0.3, 3)) for _ in range(level + 1)] if rng.integers(2) else float(rng.uniform(0.3, 3))
    scaling = bool(rng.integers(2))
    out = {}
    try:
        for mode in (1, 0):
            L.set_option("bilateral_paired", mode)
            out[mode] = W.AtrousTransform(fam, bilateral=sig, bilateral_scaling=scaling)(a, level).data.copy()
    finally:
        L.set_option("bilateral_paired", 1)
    tag = f"bilateral pairs {fam.__name__} {H}x{Wd} L={level} sigma={sig if np.isscalar(sig) else 'list'} scaling={scaling}"
    if not np.array_equal(_bits(out[0]), _bits(out[1])):
        return f"{tag}: paired loads != generic loads"
    if H * Wd <= 400000:
        ref = O.atrous_standard(a, level, "b3spline" if fam is W.B3spline else "triangle", bilateral=sig, bilateral_scaling=scaling)
        err = float(np.abs(out[1] - ref).max())
        bound = 4e-5 * float(np.abs(a).max())
        if not err <= bound:
            return f"{tag}: vs numpy oracle {err:.3e} > {bound:.3e}"
    return None


def case_fused_any_width(rng):
    f64 = bool(rng.integers(2))
    b3 = bool(rng.integers(2))
    level = int(rng.integers(2, 9 if not b3 else 7))
    H = int(rng.integers(16, 1300))
    Wd = int(rng.integers(9, 2300))
    a = rng.standard_normal((H, Wd)) * 5 + 20
    if f64:
        import wavelets_amd as WA
        taps = tuple(float(t) for t in (WA.B3spline if b3 else WA.Triangle).coefficients_1d)
        plan = L.Plan64(L.default_context(), H, Wd, taps, level)
        plan.upload(L.PLANE_INPUT, a)
    else:
        plan = L.Plan(L.default_context(), H, Wd, L.B3SPLINE if b3 else L.TRIANGLE, level)
        plan.upload(L.PLANE_INPUT, a.astype(np.float32))
    got = {}
    try:
        for mode in (1, 0):
            L.set_option("fused_fast", mode)
            if f64:
                plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT)
            else:
                plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, L.FLAG_FUSED)
            got[mode] = [plan.download(s).copy() for s in range(level + 1)] + [plan.download(L.PLANE_OUT).copy()]
    finally:
        L.set_option("fused_fast", 1)
        plan.close()
    for i, (x, y) in enumerate(zip(got[1], got[0])):
        if not np.array_equal(_bits(x), _bits(y)):
            return f"fused any width {'f64' if f64 else 'f32'} {'b3' if b3 else 'tri'} {H}x{Wd} L={level}: output {i} fast != generic"
    return None


def case_sequences(rng):
    n = int(rng.integers(3, 9))
    H, Wd = int(rng.integers(20, 400)), int(rng.integers(20, 600))
    dt = (np.float32, np.float64, np.int16)[int(rng.integers(3))]
    frames = [(rng.standard_normal((H, Wd)) * 20 + 100 * i).astype(dt) for i in range(n)]
    lanes = int(rng.integers(2, 5))
    weights = [[5, 3], [4, 2, 1], [3]][int(rng.integers(3))]
    noise = [float(rng.uniform(0.5, 30)) for _ in range(n)] if rng.integers(2) else None
    fam = (W.B3spline, W.Triangle)[int(rng.integers(2))]
    ref = [W.denoise(f, list(weights), fam, None if noise is None else noise[i]) for i, f in enumerate(frames)]
    got = W.denoise_many((f for f in frames), weights, fam, noise, lanes=lanes)
    for i, (x, y) in enumerate(zip(got, ref)):
        if x.dtype != y.dtype or not np.array_equal(_bits(x), _bits(y)):
            return f"denoise_many {n} x {H}x{Wd} {np.dtype(dt).name} lanes={lanes} noise={'given' if noise else 'MAD'}: frame {i} differs"
    if rng.integers(3) == 0 and dt != np.int16 and min(H, Wd) >= 40:
        kw = dict(denoise_coefficients=[5, 2]) if rng.integers(2) else dict(bilateral=1)
        ref = [W.wow(f, **kw)[0] for f in frames[:3]]
        got = W.wow_many(frames[:3], lanes=lanes, **kw)
        for i, (x, y) in enumerate(zip(got, ref)):
            if not np.array_equal(_bits(x[0]), _bits(y)):
                return f"wow_many {H}x{Wd} {np.dtype(dt).name} {kw}: frame {i} differs"
    return None


def case_median_counters(rng):
    f64 = bool(rng.integers(2))
    H, Wd = int(rng.integers(64, 1200)), int(rng.integers(64, 1800))
    kind = int(rng.integers(6))
    if kind == 0:
        a = np.full((H, Wd), float(rng.uniform(-5, 5)))                       # constant: every |w_0| is 0
    elif kind == 1:
        a = rng.standard_cauchy((H, Wd)) * 3                                   # heavy tails: most keys far outside the window
    elif kind == 2:
        a = rng.choice([0.0, 1.0, 1000.0], size=(H, Wd), p=[0.6, 0.39, 0.01])  # few values: massive ties
    elif kind == 3:
        a = rng.standard_normal((H, Wd)) * 1e-3 + 1e4                          # small noise on a pedestal
    elif kind == 4:
        a = np.where(rng.random((H, Wd)) < 0.5, rng.standard_normal((H, Wd)), 0.0)   # half the coefficients tiny
    else:
        a = rng.standard_normal((H, Wd)) * rng.uniform(0.1, 100)
    a = a.astype(np.float64 if f64 else np.float32)
    c = W.AtrousTransform(W.B3spline if rng.integers(2) else W.Triangle)(a, int(rng.integers(2, 5)))
    noise = c.get_noise()                                   # exact median of |w_0| behind the riding (windowed) histogram
    want = np.median(np.abs(c.data[0])) / 0.6745 / c.sigma_e[0]      # the reference's expression (ref wavelets.py:126-127), its promotion
    if not noise == want:
        return f"median {'f64' if f64 else 'f32'} {H}x{Wd} kind {kind}: {noise!r} != {want!r}"
    return None


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    fails = 0
    for case in range(n_cases):
        if only >= 0 and case != only:
            continue
        if case % 10 == 0:
            print(f"... case {case} of {n_cases}, {fails} failures so far", flush=True)
        rng = np.random.default_rng([seed, case])
        kind = case % 4
        try:
            msg = (case_bilateral_pairs, case_fused_any_width, case_sequences, case_median_counters)[kind](rng)
        except Exception as ex:            # noqa: BLE001
            msg = f"kind {kind}: raised {type(ex).__name__}: {ex}"
        if msg:
            fails += 1
            print(f"FAIL case {case}: {msg}", flush=True)
    print(f"{n_cases} cases, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
