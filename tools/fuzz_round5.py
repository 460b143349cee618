#!/usr/bin/env python3
"""Randomised differential test of the round-5 paths on the GPU box (run by tests/test_gpu_fuzz.py).

Every case draws from its own generator (seed, case), so any case can be replayed alone:

    python tools/fuzz_round5.py [n_cases] [seed] [only_case]

kinds: 0 the float64 stencil kernels (wt_stencil.h for double: smooth / squares / variance / detail planes / wow
update in its three forms) against the generic float64 engine, bitwise, random shapes / dilations / families;
1 the float64 bilateral transform (marching kernel) against the numpy oracle in float64; 2 wow() of a float64 or
integer image with random keywords against the numpy oracle; 3 the cfg5 flow (either precision) with the wow updates
and the early part of the plane sum on the side stream against the serial order, bitwise, random sizes, scale
counts and split points of the sum; 4 the circular products of the mixed-radix FFT (sides 2^a 3^b 5^c, either
precision) against numpy's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import atrous_numpy as O        # noqa: E402
import wavelets_amd as W                    # noqa: E402
from wavelets_amd import _lib as L          # noqa: E402
from wavelets_amd import utils as WU        # noqa: E402

TAPS = {"b3spline": (1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16), "triangle": (0.25, 0.5, 0.25)}
CLS = {"b3spline": W.B3spline, "triangle": W.Triangle}


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def case_stencil(rng):
    fam = ("b3spline", "triangle")[int(rng.integers(0, 2))]
    H, Wd = int(rng.integers(2, 700)), int(rng.integers(1, 1500))
    if rng.integers(0, 3) == 0:
        Wd = int(rng.integers(1, 600)) * 2                    # even widths reach the lattice kernel
    s = int(rng.integers(0, 11))
    a = rng.standard_normal((H, Wd)) * float(rng.uniform(0.1, 100)) + float(rng.uniform(-1e3, 1e3))
    nz = rng.uniform(0.5, 2.0, (H, Wd))
    tau, soft = float(rng.uniform(0, 2)) * float(np.abs(a).std()), bool(rng.integers(0, 2))
    use_nz, use_gm, root = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    ctx = L.default_context()
    res = {}
    for on in (1, 0):
        L.set_option("stencil64", on)
        try:
            p = L.Plan64(ctx, H, Wd, TAPS[fam], 1)
            try:
                A, B, NZ, GM = L.PLANE_SCRATCH(2), L.PLANE_SCRATCH(3), L.PLANE_SCRATCH(4), L.PLANE_SCRATCH(5)
                p.upload(A, a)
                p.upload(NZ, nz)
                out = []
                p.smooth(A, B, s); out.append(p.download(B))
                p.smooth(A, B, s, True); out.append(p.download(B))
                p.local_variance(A, B, s, 1.3, 2.0, root)
                out.append(p.download(B))
                p.copy(A, 0)
                p.fill(GM, 0.25)
                p.wow_scale(0, s, tau, soft, NZ if use_nz else L.PLANE_NONE, 0.7, GM if use_gm else L.PLANE_NONE)
                out += [p.download(0), p.download(GM)]
                res[on] = out
            finally:
                p.close()
        finally:
            L.set_option("stencil64", 1)
    for k in range(5):
        if not np.array_equal(bits(res[1][k]), bits(res[0][k])):
            return f"stencil64 {fam} {H}x{Wd} s={s} tau={tau:.3g} soft={soft} noise={use_nz} gamma={use_gm} sqrt={root}: output {k} differs"
    return None


def case_bilateral64(rng):
    fam = ("b3spline", "triangle")[int(rng.integers(0, 2))]
    H, Wd = int(rng.integers(2, 160)), int(rng.integers(2, 200))
    level = int(rng.integers(1, 6))
    sig = float(rng.uniform(0.3, 3)) if rng.integers(0, 2) else [float(rng.uniform(0.3, 3)) for _ in range(int(rng.integers(1, level + 2)))]
    scaling = bool(rng.integers(0, 2))
    img = rng.standard_normal((H, Wd)) * float(rng.uniform(0.5, 20)) + float(rng.uniform(-100, 100))
    if rng.integers(0, 2):
        # FITS-like integers (no wrap-around: the variance conv(I^2) - conv(I)^2 of ref:25-27 cancels, and a frame
        # that jumps between 0 and 65535 leaves ~1e-10 of it to the order of the additions - in the reference too)
        dt = (np.int16, np.int32, ">i2", np.uint16)[int(rng.integers(0, 4))]
        img = np.round(np.clip(img + (200 if dt is np.uint16 else 0), 0 if dt is np.uint16 else -30000, 30000)).astype(dt)
    got = W.AtrousTransform(CLS[fam], bilateral=sig, bilateral_scaling=scaling)(img, level).data
    ref = O.atrous_standard(np.asarray(img, np.float64), level, fam, bilateral=sig, bilateral_scaling=scaling)
    amax = max(1.0, float(np.abs(np.asarray(img, np.float64)).max()))
    e = float(np.abs(got - ref).max())
    from wavelets_amd.wavelets import _result_dtype
    want_dt = _result_dtype(img)          # (big-endian int16 is NOT among the types the reference recasts, ref:297: float32)
    if got.dtype != want_dt or not e <= (1e-12 if want_dt == np.float64 else 2e-5) * amax:
        return f"bilateral64 {fam} {H}x{Wd} {img.dtype} L={level} sigma={sig} scaling={scaling}: {got.dtype}, max err {e:.3e} (max|img| {amax:.3g})"
    return None


def case_wow64(rng):
    fam = ("b3spline", "triangle")[int(rng.integers(0, 2))]
    H, Wd = int(rng.integers(40, 200)), int(rng.integers(40, 260))
    img = rng.standard_normal((H, Wd)) * 5 + 3 * np.sin(np.arange(Wd) / 9.0)[None, :] + 40.0
    if rng.integers(0, 3) == 0:
        img = np.round(img * 10).astype(np.int16)
    kw = {}
    if rng.integers(0, 2):
        kw["bilateral"] = 1 if rng.integers(0, 2) else [1.5, 1.0]
    if rng.integers(0, 2):
        kw["denoise_coefficients"] = [5, 2][: int(rng.integers(1, 3))]
    if rng.integers(0, 4) == 0:
        kw["h"], kw["gamma"] = 0.5, 2.0
        kw.setdefault("denoise_coefficients", [5, 2])
    if rng.integers(0, 4) == 0:
        kw["preserve_variance"] = True
    if rng.integers(0, 4) == 0:
        kw["n_scales"] = int(rng.integers(2, 5))
    rec, co = W.wow(img.copy(), CLS[fam], **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    rref, cref = O.wow(np.asarray(img, np.float64).copy(), fam, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    if co.data.shape != cref.data.shape:
        return f"wow64 {fam} {H}x{Wd} {kw}: plane stack {co.data.shape} != {cref.data.shape}"
    e_i = float(np.abs(rec - rref).max()) / max(1.0, float(np.abs(rref).max()))
    e_p = float(np.abs(co.data - cref.data).max()) / max(1.0, float(np.abs(cref.data).max()))
    if rec.dtype != np.float64 or not (e_i <= 2e-11 and e_p <= 2e-11):
        return f"wow64 {fam} {H}x{Wd} {img.dtype} {kw}: {rec.dtype}, image {e_i:.3e} planes {e_p:.3e}"
    return None


def case_side_stream(rng):
    fam = ("b3spline", "triangle")[int(rng.integers(0, 2))]
    H, Wd = int(rng.integers(64, 900)), int(rng.integers(64, 1200))
    f64 = bool(rng.integers(0, 2))
    level = int(rng.integers(1, max(2, int(np.log2(min(H, Wd) / len(TAPS[fam]))))))
    img = (rng.standard_normal((H, Wd)) + 3 * np.sin(np.arange(Wd) / 30.0)[None, :]).astype(np.float64 if f64 else np.float32)
    dc = [5, 2][: int(rng.integers(0, 3))]
    ctx = L.default_context()
    res = {}
    keep_tail = WU._SUM_TAIL_PLANES, WU._SUM_TAIL_PLANES_F64
    WU._SUM_TAIL_PLANES = WU._SUM_TAIL_PLANES_F64 = int(rng.integers(1, 4))     # (the early sum starts at 2 x tail planes)
    for on in (1, 0):
        L.set_option("wow_overlap", on)
        try:
            sb = [1] * (level + 1)
            tr = W.AtrousTransform(CLS[fam], bilateral=sb)
            plan = L.Plan64(ctx, H, Wd, TAPS[fam], level) if f64 else L.Plan(ctx, H, Wd, {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam], level)
            try:
                plan.upload(L.PLANE_INPUT, img)
                co = W.Coefficients(plan, CLS[fam](2), sb)
                for _ in range(2):
                    tr._run(plan, level)
                    co.noise = None
                    WU._wow_device(co, level, [], True, list(dc), True, False, 3.2, None, None, 0)
                res[on] = [plan.download(s) for s in range(level + 1)] + [plan.download(L.PLANE_OUT)]
                co._plan = None
            finally:
                plan.close()
        finally:
            L.set_option("wow_overlap", 1)
            if on == 0:
                WU._SUM_TAIL_PLANES, WU._SUM_TAIL_PLANES_F64 = keep_tail
    for k, (u, v) in enumerate(zip(res[1], res[0])):
        if not np.array_equal(bits(u), bits(v)):
            return f"side stream {fam} {H}x{Wd} {'f64' if f64 else 'f32'} L={level} dc={dc}: output {k} differs"
    return None


def case_fft(rng):
    def side():
        while True:
            n = 2 ** int(rng.integers(0, 8)) * 3 ** int(rng.integers(0, 5)) * 5 ** int(rng.integers(0, 4))
            if 2 <= n <= 2400:
                return n
    H, Wd = side(), side()
    f64 = bool(rng.integers(0, 2))
    dt = np.float64 if f64 else np.float32
    x = rng.standard_normal((H, Wd)).astype(dt)
    k = np.zeros((H, Wd), dt)
    kh, kw = int(rng.integers(1, min(H, 12) + 1)), int(rng.integers(1, min(Wd, 12) + 1))
    k[:kh, :kw] = rng.random((kh, kw))
    k /= k.sum()
    k = np.roll(k, (-(kh // 2), -(kw // 2)), axis=(0, 1))
    f = np.fft.fft2(k.astype(np.float64))
    X = np.fft.fft2(x.astype(np.float64))
    want = (np.fft.ifft2(X * f).real, np.fft.ifft2(X * f.conj()).real)
    if not L.fft_supported(H, Wd):
        return f"fft {H}x{Wd}: not supported"
    ctx = L.default_context()
    plan = L.Plan64(ctx, H, Wd, TAPS["b3spline"], 0) if f64 else L.Plan(ctx, H, Wd, L.B3SPLINE, 0)
    try:
        S = L.PLANE_SCRATCH(6)
        plan.upload(S, k)
        plan.upload(L.PLANE_INPUT, x)
        plan.fft_spectrum(S)
        bound = (6e-6 if not f64 else 2e-13) * float(np.abs(x).max())
        for conj in (False, True):
            plan.fft_apply(L.PLANE_INPUT, L.PLANE_OUT, conj)
            d = float(np.abs(plan.download(L.PLANE_OUT) - want[int(conj)]).max())
            if not d <= bound:
                return f"fft {H}x{Wd} {'f64' if f64 else 'f32'} psf {kh}x{kw} conj={conj}: max abs diff {d:.3e} > {bound:.3e}"
    finally:
        plan.close()
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
        kind = case % 5
        try:
            msg = (case_stencil, case_bilateral64, case_wow64, case_side_stream, case_fft)[kind](rng)
        except Exception as ex:            # noqa: BLE001
            msg = f"kind {kind}: raised {type(ex).__name__}: {ex}"
        if msg:
            fails += 1
            print(f"FAIL case {case}: {msg}", flush=True)
    print(f"{n_cases} cases, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
