#!/usr/bin/env python3
"""Randomised differential test of the round-4 paths on the GPU box (minutes; not part of pytest).

Every case draws from its own generator (seed, case), so any case can be replayed alone:

    python tools/fuzz_round4.py [n_cases] [seed] [only_case]

  kind 0  denoise() float32: the riding WINDOWED histogram behind the first fused pass, the threshold step
          between the passes - random shape / family / thresholds / data distribution (Gaussian, offset,
          heavy-tailed, quantised, nearly constant) against the numpy oracle; the noise estimate must be
          the exact np.median of the product's own plane 0
  kind 1  the same on the float64 engine (1e-12)
  kind 2  richardson_lucy(fft=True): power-of-two images (the engine's FFT) and other sizes (extended
          frame), odd heights, odd / even PSFs, float32 / float64, with the tap threshold forced to 1 -
          against the oracle's direct periodic form
  kind 3  atrous_convolution with every np.pad mode, 1-D / 2-D / 3-D, any kernel shape, dilation 2^s
  kind 4  scaling functions with an even number of taps or more than 15: standard / recursive transforms,
          1-D / 2-D, against the oracle's tap-list restatement
  kind 5  integer / big-endian images through denoise(): the device widening against the oracle on the
          promoted array (float64 for the reference's recast list, float32 otherwise)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import atrous_numpy as O        # noqa: E402
import wavelets_amd as W                    # noqa: E402
from wavelets_amd import utils as WU        # noqa: E402

PAD_MODES = ["symmetric", "reflect", "edge", "wrap", "constant", "maximum", "minimum", "mean", "median", "linear_ramp", "empty"]


def draw_image(rng, shape, dtype):
    kind = int(rng.integers(0, 5))
    a = rng.standard_normal(shape)
    if kind == 1:
        a = a * float(rng.uniform(0.01, 50)) + float(rng.uniform(-1e3, 1e3))
    elif kind == 2:
        a = np.clip(rng.standard_cauchy(shape), -1e4, 1e4)
    elif kind == 3:
        a = np.round(a * 4) / 4 + 10
    elif kind == 4:
        a = np.full(shape, 3.0) + (rng.random(shape) < 0.02) * rng.standard_normal(shape)
    return a.astype(dtype), ("gauss", "offset", "cauchy", "quantised", "nearly constant")[kind]


def case_denoise(rng, f64):
    H, Wd = int(rng.integers(4, 1400)), int(rng.integers(4, 2200))
    cls, fam = ((W.B3spline, "b3spline"), (W.Triangle, "triangle"))[int(rng.integers(0, 2))]
    level = int(rng.integers(1, 9))
    sig = [float(rng.choice([0, 0, 1, 2, 3, 5])) for _ in range(level)]
    soft = bool(rng.integers(0, 2))
    dtype = np.float64 if f64 else np.float32
    a, what = draw_image(rng, (H, Wd), dtype)
    tag = f"denoise {dtype.__name__} {H}x{Wd} {fam} sigma={sig} soft={soft} data={what}"
    got = W.denoise(a.copy(), sig, scaling_function=cls, soft_threshold=soft)
    want = O.denoise(a.copy(), sig, family=fam, soft_threshold=soft)
    scale = max(1.0, float(np.abs(a).max()))
    tol = (1e-11 if f64 else 2e-5) * scale
    if not soft:
        # A hard threshold flips when a coefficient sits within rounding distance of it (the two sides
        # round the transform differently in the last bits - and data on a large offset is coarsely
        # quantised, so MANY coefficients sit there).  Exact check: pixels where some plane's |w| is
        # within `amb` of its threshold are set aside (and must stay a minority); all others must agree.
        c = O.Coeffs(O.atrous_standard_nd(a.copy(), level, fam), fam)
        noise = c.get_noise()
        amb = (1e-13 if f64 else 4e-6) * scale
        ambiguous = np.zeros(a.shape, dtype=bool)
        for scl, sg in enumerate(sig):
            if sg != 0 and noise != 0:
                ambiguous |= np.abs(np.abs(c.data[scl]) - sg * noise * c.sigma_e[scl]) <= amb
        bad = (np.abs(got - want) > tol) & ~ambiguous
        ok = not bad.any() and ambiguous.mean() < 0.5       # (heavy tails: the band scales with max|a|)
        if not ok:
            return tag + f": {int(bad.sum())} pixels differ away from a threshold ({100 * ambiguous.mean():.2f} % ambiguous)"
    else:
        ok = bool(np.abs(got - want).max() <= tol)
    if got.dtype != dtype:
        return tag + f": dtype {got.dtype}"
    if not ok:
        return tag + f": max err {np.abs(got - want).max():.3e} tol {tol:.1e}"
    # the noise estimate: exact median of the product's own plane 0
    c = W.AtrousTransform(cls)(a.copy(), level)
    n = c.get_noise()
    w0 = np.asarray(c.data[0])
    ref = np.median(np.abs(w0)) / 0.6745 / c.sigma_e[0]
    if not np.isclose(n, ref, rtol=1e-15 if f64 else 1e-6, atol=0):
        return tag + f": noise {n!r} vs {ref!r}"
    return None


def case_rl(rng):
    f64 = bool(rng.integers(0, 2))
    pow2 = bool(rng.integers(0, 2))
    if pow2:
        H, Wd = int(2 ** rng.integers(4, 8)), int(2 ** rng.integers(4, 8))
    else:
        H, Wd = int(rng.integers(12, 150)), int(rng.integers(6, 90)) * 2
    kh, kw = int(rng.integers(1, min(H, 30) + 1)), int(rng.integers(1, min(Wd, 30) + 1))
    yy, xx = np.mgrid[0:H, 0:Wd]
    truth = np.exp(-((yy - H / 3.) ** 2 + (xx - Wd / 2.) ** 2) / 40.) * 30 + 2.0
    data = truth + rng.standard_normal(truth.shape) * 0.3
    psf = rng.uniform(0.1, 1.0, (kh, kw)) * np.hanning(kh + 2)[1:-1, None] * np.hanning(kw + 2)[None, 1:-1]
    psf /= psf.sum()
    ft = np.float64 if f64 else np.float32
    kw_args = dict(iterations=int(rng.integers(1, 4)), fft=True)
    tag = f"richardson_lucy {ft.__name__} {H}x{Wd} psf {kh}x{kw} {kw_args}"
    keep, WU._FFT_MIN_TAPS = WU._FFT_MIN_TAPS, 1
    try:
        got = W.richardson_lucy(data.astype(ft), psf.astype(ft), **kw_args)
    finally:
        WU._FFT_MIN_TAPS = keep
    want = O.richardson_lucy(data.astype(ft), psf.astype(ft), **kw_args)
    tol = (1e-9 if f64 else 3e-4) * float(np.abs(want).max())
    err = float(np.abs(got - want).max())
    return None if err <= tol else tag + f": max err {err:.3e} tol {tol:.1e}"


def case_pad(rng):
    nd = int(rng.integers(1, 4))
    shp = {1: (int(rng.integers(9, 400)),), 2: (int(rng.integers(6, 90)), int(rng.integers(6, 120))),
           3: (int(rng.integers(4, 9)), int(rng.integers(5, 20)), int(rng.integers(5, 30)))}[nd]
    s = int(rng.integers(0, 3))
    ks = tuple(int(rng.integers(1, 6)) for _ in range(nd))
    mode = PAD_MODES[int(rng.integers(0, len(PAD_MODES)))]
    f64 = bool(rng.integers(0, 2))
    ft = np.float64 if f64 else np.float32
    img = rng.standard_normal(shp).astype(ft)
    ker = rng.random(ks).astype(ft)
    tag = f"atrous_convolution {ft.__name__} {shp} kernel {ks} mode={mode} s={s}"
    if mode == "empty":
        return None                        # np.pad leaves the border uninitialised: nothing to compare
    got = W.atrous_convolution(img, ker, s=s, mode=mode)
    want = O.atrous_convolution_nd(img, ker, None, s, mode)
    tol = (1e-12 if f64 else 3e-6) * max(1.0, float(np.abs(want).max()))
    err = float(np.abs(np.asarray(got) - want).max())
    return None if err <= tol else tag + f": max err {err:.3e} tol {tol:.1e}"


def make_sf(taps):
    from wavelets_amd.wavelets import AbstractScalingFunction

    class Fz(AbstractScalingFunction):
        coefficients_1d = np.asarray(taps)
        sigma_e_1d = sigma_e_2d = sigma_e_3d = np.ones(12)

        def __init__(self, *args, **kwargs):
            super().__init__("fuzz", *args, **kwargs)
    return Fz


def case_generic(rng):
    n = int(rng.choice([2, 4, 6, 8, 16, 17, 19]))
    taps = rng.uniform(0.2, 1.0, n)
    taps = taps / taps.sum()
    nd = int(rng.integers(1, 3))
    shp = (int(rng.integers(40, 400)),) if nd == 1 else (int(rng.integers(20, 120)), int(rng.integers(20, 160)))
    level = int(rng.integers(1, 4))
    a = rng.standard_normal(shp).astype(np.float32)
    mode = int(rng.integers(0, 2))
    tag = f"generic taps n={n} {shp} L={level} {'recursive' if mode else 'standard'}"
    sf = make_sf(taps.astype(np.float64))
    try:
        c = W.AtrousTransform(sf)(a.copy(), level, recursive=bool(mode))
    except Exception as ex:                # noqa: BLE001
        return tag + f": raised {type(ex).__name__}: {ex}"
    want = (O.atrous_recursive_taps_nd if mode else O.atrous_standard_taps_nd)(a, level, taps)
    err = float(np.abs(np.asarray(c.data) - want).max())
    tol = 2e-5 * max(1.0, float(np.abs(a).max()))
    return None if err <= tol else tag + f": max err {err:.3e} tol {tol:.1e}"


def case_elem_types(rng):
    """integer / byte-swapped inputs through denoise(): the device widening (wt_upload_int / wt64_upload_int)
    against the oracle on the array promoted the way the reference (or, for types it does not recast,
    this engine) promotes it"""
    H, Wd = int(rng.integers(8, 700)), int(rng.integers(8, 900))
    dts = ["u1", "i1", "<i2", "<u2", "<i4", "<u4", "<i8", ">i2", ">u2", ">i4", ">f4", ">f8"]
    dt = np.dtype(dts[int(rng.integers(0, len(dts)))])
    if dt.kind == "f":
        a = (rng.standard_normal((H, Wd)) * 30 + 100).astype(dt)
    else:
        info = np.iinfo(dt)
        lo, hi = max(info.min, -30000), min(info.max, 30000)
        a = np.clip(rng.standard_normal((H, Wd)) * (hi - lo) / 12 + (hi + lo) / 2, lo, hi).astype(dt)
    cls, fam = ((W.B3spline, "b3spline"), (W.Triangle, "triangle"))[int(rng.integers(0, 2))]
    level = int(rng.integers(1, 6))
    sig = [float(rng.choice([0, 1, 2, 3])) for _ in range(level)]
    tag = f"denoise {dt.str} {H}x{Wd} {fam} sigma={sig}"
    got = W.denoise(a, sig, scaling_function=cls)
    recast = dt in [np.dtype(t) for t in (np.int32, np.int64, '>f4', '>f8', 'int16', 'uint16', 'int32', 'uint32')]
    promoted = a.astype(np.float64 if recast else np.float32)
    want = O.denoise(promoted.copy(), sig, family=fam)
    if got.dtype != promoted.dtype:
        return tag + f": dtype {got.dtype}, expected {promoted.dtype}"
    tol = (1e-11 if recast else 2e-5) * max(1.0, float(np.abs(promoted).max()))
    err = float(np.abs(got - want).max())
    return None if err <= tol else tag + f": max err {err:.3e} tol {tol:.1e}"


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
        kind = case % 6
        try:
            msg = (lambda: case_denoise(rng, False), lambda: case_denoise(rng, True), lambda: case_rl(rng),
                   lambda: case_pad(rng), lambda: case_generic(rng), lambda: case_elem_types(rng))[kind]()
        except Exception as ex:            # noqa: BLE001
            msg = f"kind {kind}: raised {type(ex).__name__}: {ex}"
        if msg:
            fails += 1
            print(f"FAIL case {case}: {msg}", flush=True)
    print(f"{n_cases} cases, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
