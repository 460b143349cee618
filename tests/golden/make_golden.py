#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference/watroo) in the build container.  Not product code; never runs on the GPU box.

    python tests/golden/make_golden.py            # python3.10 / numpy 2.x (fixtures g*.npz)
    /opt/conda/bin/python3.9 tests/golden/make_golden.py --real-numexpr
                                                  # numpy 1.26 + REAL numexpr 2.7.3 -> g5_realne.npz

The reference imports two third-party wheels that are absent from this image
(requirements.txt:2,4): ``cv2`` (opencv-python) and - under python3.10 - ``numexpr``.
Stand-ins are registered in ``sys.modules`` *before* importing it:

* ``cv2.filter2D`` -> OpenCV's documented semantics for the only call form watroo uses
  (wavelets.py:39-45: ddepth=-1, anchor=(-1,-1), delta=0, BORDER_REFLECT): correlation with
  the kernel, centre anchor, border ``fedcba|abcdefgh|hgfedcb`` == scipy.ndimage 'reflect'.
  This pins semantics, not OpenCV's rounding.  Fixture groups that depend on it are tagged
  ``pin = "semantic(cv2 stand-in)"``.
* ``numexpr.evaluate`` -> evaluates the single expression of wavelets.py:97 with numpy in
  the caller's frame.  ``--real-numexpr`` regenerates the bilateral group with the real
  numexpr under /opt/conda's interpreter to show the stand-in is faithful.

Fixture groups that never touch cv2 (the reference's own pure-numpy ``atrous_convolution``,
``Coefficients`` methods, ``generalized_anscombe``) are tagged ``pin = "hard"``.
"""
import os
import sys
import types

import numpy as np
import scipy
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
REAL_NE = "--real-numexpr" in sys.argv


def _install_standins():
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_REFLECT = 2
    cv2.BORDER_REFLECT_101 = 4

    def filter2D(src, ddepth, kernel, dst=None, anchor=(-1, -1), delta=0, borderType=4):
        assert ddepth == -1 and tuple(anchor) == (-1, -1)
        mode = {2: "reflect", 4: "mirror"}[borderType]
        k = np.asarray(kernel)
        k = k[None, :] if k.ndim == 1 else k
        res = ndimage.correlate(np.asarray(src), k.astype(src.dtype), mode=mode) + delta
        if dst is not None:
            dst[...] = res
            return dst
        return res

    cv2.filter2D = filter2D
    sys.modules["cv2"] = cv2
    try:
        import numexpr  # noqa: F401  (real one, conda python)
        return "real numexpr " + numexpr.__version__
    except ImportError:
        ne = types.ModuleType("numexpr")

        def evaluate(expr, out=None, **kw):
            f = sys._getframe(1)
            ns = {**f.f_globals, **f.f_locals, "exp": np.exp, "sqrt": np.sqrt}
            r = eval(expr, {}, ns)
            if out is not None:
                out[...] = r
                return out
            return r

        ne.evaluate = evaluate
        sys.modules["numexpr"] = ne
        return "numexpr stand-in (numpy eval)"


NE_KIND = _install_standins()
sys.path.insert(0, "/root/reference")
import watroo  # noqa: E402
from watroo import (AtrousTransform, B3spline, Triangle, Coefficients,  # noqa: E402
                    generalized_anscombe, convolution, denoise, wow, richardson_lucy)
from watroo.wavelets import atrous_convolution, sdev_loc  # noqa: E402

FAM = {"b3spline": B3spline, "triangle": Triangle}
META = dict(numpy=np.__version__, scipy=scipy.__version__, watroo=watroo.__version__,
            numexpr=NE_KIND)


def img(shape, seed, positive=False):
    a = np.random.default_rng(seed).standard_normal(shape).astype(np.float32)
    if positive:
        a = (np.abs(a) * 20 + 5).astype(np.float32)
    return a


def save(name, pin, **arrays):
    arrays["_meta"] = np.array(repr(dict(META, pin=pin)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  pin={pin}")


SHAPES = [((37, 53), 1), ((64, 48), 2), ((16, 16), 3)]


def g0_hard():
    """Reference's own numpy operator + pointwise code: no cv2 anywhere in the arithmetic.
    To prove it, the stand-in's filter2D is replaced by a function that raises while this
    group is generated (the `cv2` module only has to EXIST for `import watroo` to succeed)."""
    import cv2 as _cv2
    _real = _cv2.filter2D

    def _forbidden(*a, **k):
        raise AssertionError("hard-pin group must not call cv2.filter2D")
    _cv2.filter2D = _forbidden
    try:
        _g0_hard_body()
    finally:
        _cv2.filter2D = _real


def _g0_hard_body():
    out = {}
    for shape, seed in SHAPES:
        a = img(shape, seed)
        tag = f"{shape[0]}x{shape[1]}"
        out[f"img_{tag}"] = a
        for fam, cls in FAM.items():
            k = cls(2).kernel.astype(a.dtype)
            for s in range(5):
                out[f"aconv_{fam}_{tag}_s{s}"] = atrous_convolution(a, k, None, s, "symmetric")
    # Coefficients methods on a synthetic stack (data need not come from a transform)
    stack = img((4, 37, 53), 11) * np.array([1, .3, .1, 2], np.float32)[:, None, None]
    out["stack"] = stack.copy()
    for fam, cls in FAM.items():
        c = Coefficients(stack.copy(), cls(2))
        out[f"noise_{fam}"] = np.float64(c.get_noise())
        out[f"sig_hard_3_2_{fam}"] = c.significance(3, 2, soft_threshold=False)
        out[f"sig_soft_3_1_{fam}"] = c.significance(3, 1)
        c = Coefficients(stack.copy(), cls(2)); c.denoise([5, 3])
        out[f"den_53_{fam}"] = c.data
        c = Coefficients(stack.copy(), cls(2)); c.denoise([5, 3, 2], weights=[.5, 2, 1])
        out[f"den_532_w_{fam}"] = c.data
        c = Coefficients(stack.copy(), cls(2)); c.denoise([5, 3], soft_threshold=False)
        out[f"den_53_hard_{fam}"] = c.data
        c = Coefficients(stack.copy(), cls(2)); c.noise = 0.7; c.denoise([3, 2])
        out[f"den_32_noise07_{fam}"] = c.data
        nmap = (np.abs(img((37, 53), 12)) + .5).astype(np.float32)
        out["noise_map"] = nmap
        c = Coefficients(stack.copy(), cls(2)); c.noise = nmap; c.denoise([3, 2])
        out[f"den_32_noisemap_{fam}"] = c.data
        c = Coefficients(stack.copy(), cls(2), bilateral=[1, 1]); c.denoise([5, 3])
        out[f"den_53_bilat_{fam}"] = c.data
    cz = Coefficients(np.zeros((3, 8, 8), np.float32), B3spline(2)); cz.denoise([5, 3])
    out["den_zero_noise_branch"] = cz.data
    out["noise_zero"] = np.float64(cz.noise)
    p = img((37, 53), 13, positive=True)
    out["ans_in"] = p
    out["ans_fwd"] = generalized_anscombe(p)
    out["ans_fwd_params"] = generalized_anscombe(p, alpha=2., g=1., sigma=.5)
    out["ans_inv"] = generalized_anscombe(generalized_anscombe(p), inverse=True)
    save("g0_hard", "hard", **out)


def g1_transform():
    out = {}
    for shape, seed in SHAPES:
        a = img(shape, seed)
        tag = f"{shape[0]}x{shape[1]}"
        out[f"img_{tag}"] = a
        for fam, cls in FAM.items():
            for L in (1, 2, 3, 4, 5):
                out[f"coef_{fam}_{tag}_L{L}"] = AtrousTransform(cls)(a, L).data
            out[f"conv_{fam}_{tag}_s2"] = convolution(a, cls(2), s=2)
    save("g1_transform", "semantic(cv2 stand-in)", **out)


def g2_g3_denoise():
    out = {}
    a = img((64, 48), 2)
    out["img"] = a
    for fam, cls in FAM.items():
        c = AtrousTransform(cls)(a, 4)
        out[f"noise_{fam}"] = np.float64(c.get_noise())
        c.denoise([5, 3])
        out[f"coef_den_53_{fam}"] = c.data
        out[f"denoise_53_{fam}"] = denoise(a, [5, 3], cls)
        out[f"denoise_532_hard_{fam}"] = denoise(a, [5, 3, 2], cls, soft_threshold=False)
        out[f"denoise_53_noise_{fam}"] = denoise(a, [5, 3], cls, noise=0.9)
    p = img((64, 48), 5, positive=True)
    out["img_pos"] = p
    out["denoise_53_anscombe"] = denoise(p, [5, 3], Triangle, anscombe=True)
    save("g2_denoise", "semantic(cv2 stand-in)", **out)


def g4_wow():
    out = {}
    a = (img((64, 64), 4) + 3 * np.sin(np.arange(64) / 5.)[None, :]).astype(np.float32)
    out["img"] = a
    cases = {
        "default": dict(),
        "triangle": dict(scaling_function=Triangle),
        "dc52": dict(denoise_coefficients=[5, 2]),
        "n3_w_dc": dict(n_scales=3, weights=[.5], denoise_coefficients=[5, 2]),
        "h05_g2": dict(h=.5, gamma=2, denoise_coefficients=[5, 2]),
        "h1": dict(h=1, denoise_coefficients=[5, 2]),
        "pv": dict(preserve_variance=True, denoise_coefficients=[5, 2]),
        "nowhite": dict(whitening=False, denoise_coefficients=[5, 2]),
        "hard": dict(denoise_coefficients=[5, 2], soft_threshold=False),
        "bilat1": dict(bilateral=1),
        "bilat1_dc52": dict(bilateral=1, denoise_coefficients=[5, 2]),
        "bilat_list_scaling": dict(bilateral=[1.5, 1.], bilateral_scaling=True,
                                   denoise_coefficients=[4]),
    }
    for name, kw in cases.items():
        recon, coef = wow(a.copy(), **kw)
        out[f"recon_{name}"] = recon
        out[f"coef_{name}"] = coef.data
        out[f"noise_{name}"] = np.float64(np.nan if coef.noise is None else coef.noise)
    # Coefficients input (utils.py:128-131, 152-153): same object returned
    c = AtrousTransform()(a.copy(), 3)
    recon, c2 = wow(c, denoise_coefficients=[5, 2])
    assert c2 is c
    out["recon_from_coeffs"] = recon
    out["coef_from_coeffs"] = c.data
    save("g4_wow", "semantic(cv2 stand-in)", **out)


def g5_bilateral(name="g5_bilateral"):
    out = {}
    a = img((37, 53), 6)
    out["img"] = a
    for fam, cls in FAM.items():
        sf = cls(2)
        for s in (0, 1, 2):
            var = sdev_loc(a, sf, s=s, variance=True)
            out[f"var_{fam}_s{s}"] = var
            out[f"sdev_{fam}_s{s}"] = sdev_loc(a, sf, s=s)
            out[f"bconv_{fam}_s{s}"] = atrous_convolution(a, sf.kernel.astype(a.dtype), var, s,
                                                          "symmetric")
        out[f"coef_b1_{fam}"] = AtrousTransform(cls, bilateral=1)(a, 3).data
        out[f"coef_blist_scaling_{fam}"] = AtrousTransform(
            cls, bilateral=[2., .5], bilateral_scaling=True)(a, 3).data
    save(name, "semantic(cv2 stand-in); numexpr: " + NE_KIND, **out)


def g7_recursive_g8_tests():
    out = {}
    a = img((64, 48), 2)
    out["img"] = a
    out["recursive_b3_L3"] = AtrousTransform()(a, 3, recursive=True).data
    ones = np.ones((128, 128))
    out["ones_L4"] = AtrousTransform()(ones, 4).data          # tests/test_wavelets.py:8-13
    r, c = wow(ones)                                            # tests/test_utils.py:7-9
    out["wow_ones"] = r
    r, c = wow(ones, bilateral=True)
    out["wow_ones_bilateral"] = r
    # integer input recast (wavelets.py:297,319-320)
    ai = (img((16, 16), 9) * 100).astype(np.int32)
    out["img_int32"] = ai
    out["coef_int32_L2"] = AtrousTransform()(ai, 2).data
    save("g7_misc", "semantic(cv2 stand-in)", **out)


def g9_richardson_lucy():
    """SURVEY 8f rank 1.  Positive image blurred by a small PSF + noise; fft=False branch."""
    out = {}
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[:48, :40]
    truth = (5 + 40 * np.exp(-((yy - 20) ** 2 + (xx - 15) ** 2) / 18.)
             + 25 * np.exp(-((yy - 33) ** 2 + (xx - 28) ** 2) / 8.)).astype(np.float32)
    g = np.exp(-(np.arange(-2, 3) ** 2) / 2.0)
    psf = np.outer(g, g * np.array([1, 1, 1, .8, .6]))      # 5x5, deliberately asymmetric
    psf = (psf / psf.sum()).astype(np.float32)
    import cv2
    blurred = cv2.filter2D(truth, -1, psf[::-1, ::-1], None, (-1, -1), 0, cv2.BORDER_REFLECT)
    data = (blurred + rng.standard_normal(truth.shape) * .5).astype(np.float32)
    out["data"], out["psf"] = data, psf
    cases = {
        "soft": dict(iterations=3),
        "hard": dict(iterations=3, threshold_type='hard'),
        "uniform": dict(iterations=2, uniform_init=True),
        "soft_nonpersistent": dict(iterations=3, persistent_mrs=False, denoise_coefficients=(4, 2)),
        "hard_nonpersistent": dict(iterations=2, threshold_type='hard', persistent_mrs=False),
    }
    for name, kw in cases.items():
        out[f"rl_{name}"] = richardson_lucy(data.copy(), psf, **kw)
    even = np.ones((4, 6), np.float32) / 24
    out["psf_even"] = even
    out["filter_even"] = cv2.filter2D(data, -1, even, None, (-1, -1), 0, cv2.BORDER_REFLECT)
    save("g9_richardson_lucy", "semantic(cv2 stand-in)", **out)


def g13_richardson_lucy_fft():
    """richardson_lucy(fft=True): circular rfft2 products (numpy only, no cv2 in the loop)."""
    g9 = np.load(os.path.join(HERE, "g9_richardson_lucy.npz"))
    data, psf = g9["data"], g9["psf"]
    out = {"data": data, "psf": psf}
    rng = np.random.default_rng(77)
    even = rng.uniform(0.2, 1.0, (4, 6)).astype(np.float32)     # even-sized, asymmetric PSF
    even /= even.sum()
    out["psf_even"] = even
    out["rl_fft_soft"] = richardson_lucy(data.copy(), psf, iterations=3, fft=True)
    out["rl_fft_hard"] = richardson_lucy(data.copy(), psf, iterations=2, threshold_type='hard', fft=True)
    out["rl_fft_even"] = richardson_lucy(data.copy(), even, iterations=2, fft=True,
                                         denoise_coefficients=(4, 2))
    # the two circular products on their own (pins anchor / roll conventions for the even PSF)
    pad = np.zeros_like(data)
    H, W = data.shape
    kh, kw = even.shape
    pad[H // 2 - kh // 2:H // 2 - kh // 2 + kh, W // 2 - kw // 2:W // 2 - kw // 2 + kw] = even
    f = np.fft.rfft2(np.roll(pad, (H // 2, W // 2), axis=(0, 1)))
    out["circ_conv_even"] = np.fft.irfft2(np.fft.rfft2(data) * f)
    out["circ_corr_even"] = np.fft.irfft2(np.fft.rfft2(data) * f.conj())
    save("g13_rl_fft", "hard(numpy fft; transform via cv2 stand-in)", **out)


def g17_rl_fft_odd_height():
    """richardson_lucy(fft=True) on an ODD-height image: the reference rolls the padded PSF by
    H // 2 twice (utils.py:246-250), which for odd H leaves the PSF centre one row above the
    origin - the circular products are shifted by one row against the even-height case.
    (Odd widths are not valid in the reference: irfft2 returns W - 1 columns.)"""
    rng = np.random.default_rng(1717)
    data = (rng.uniform(0.5, 1.5, (33, 40)) + 4 * np.exp(-((np.arange(40) - 17.) ** 2) / 18.)[None, :]).astype(np.float32)
    psf = rng.uniform(0.2, 1.0, (5, 7)).astype(np.float32)
    psf /= psf.sum()
    thin = rng.uniform(0.2, 1.0, (1, 4)).astype(np.float32)       # anchor falls outside a 1-row PSF
    thin /= thin.sum()
    out = {"data": data, "psf": psf, "psf_thin": thin}
    out["rl_fft_odd"] = richardson_lucy(data.copy(), psf, iterations=3, fft=True, denoise_coefficients=(4, 2))
    out["rl_fft_odd_thin"] = richardson_lucy(data.copy(), thin, iterations=2, fft=True, denoise_coefficients=(4, 2))
    H, W = data.shape
    for tag, k in (("", psf), ("_thin", thin)):
        pad = np.zeros_like(data)
        kh, kw = k.shape
        pad[H // 2 - kh // 2:H // 2 - kh // 2 + kh, W // 2 - kw // 2:W // 2 - kw // 2 + kw] = k
        f = np.fft.rfft2(np.roll(pad, (H // 2, W // 2), axis=(0, 1)))
        out["circ_conv" + tag] = np.fft.irfft2(np.fft.rfft2(data) * f)
        out["circ_corr" + tag] = np.fft.irfft2(np.fft.rfft2(data) * f.conj())
    save("g17_rl_fft_odd", "hard(numpy fft; transform via cv2 stand-in)", **out)


def g18_recursive_nd_bilateral():
    """recursive=True beyond the 2-D plain case (wavelets.py:330-406): with bilateral filtering,
    on 1-D signals (scipy 'mirror' border inside every sub-array) and on cubes."""
    out = {}
    a2 = img((40, 52), 181)
    out["img2"] = a2
    out["rec2_b1"] = AtrousTransform(B3spline, bilateral=1)(a2, 3, recursive=True).data
    out["rec2_blist"] = AtrousTransform(Triangle, bilateral=[1.5, .7], bilateral_scaling=True)(a2, 3, recursive=True).data
    a1 = img((1, 200), 182)[0]
    out["sig1"] = a1
    out["rec1_b3"] = AtrousTransform(B3spline)(a1, 4, recursive=True).data
    out["rec1_tri"] = AtrousTransform(Triangle)(a1, 3, recursive=True).data
    out["rec1_b1"] = AtrousTransform(B3spline, bilateral=1)(a1, 3, recursive=True).data
    a3 = np.random.default_rng(183).standard_normal((10, 12, 14)).astype(np.float32)
    out["cube"] = a3
    out["rec3_tri"] = AtrousTransform(Triangle)(a3, 2, recursive=True).data
    out["rec3_b3"] = AtrousTransform(B3spline)(a3, 2, recursive=True).data
    out["rec3_b1"] = AtrousTransform(Triangle, bilateral=1)(a3, 2, recursive=True).data
    save("g18_recursive_nd", "1-D plain: hard (numpy + scipy); others: semantic(cv2 stand-in)", **out)


def g19_custom_bilateral_nd():
    """User-defined scaling functions beyond the plain 1-D / 2-D transform: bilateral filtering
    (2-D, 1-D, 3-D), 3-D cubes, sdev_loc, atrous_convolution with the class's own kernel and the
    recursive algorithm.  The asymmetric taps pin every orientation: cv2.filter2D correlates
    (wavelets.py:39-63), scipy convolves (:65-69), the tap loop of atrous_convolution convolves
    (:87-91)."""
    from watroo.wavelets import AbstractScalingFunction, sdev_loc, atrous_convolution

    class Binomial7(AbstractScalingFunction):
        coefficients_1d = np.array([1, 6, 15, 20, 15, 6, 1]) / 64

        def __init__(self, *args, **kwargs):
            super().__init__('binomial7', *args, **kwargs)

    class Skew5(AbstractScalingFunction):
        coefficients_1d = np.array([0.05, 0.25, 0.4, 0.2, 0.1])

        def __init__(self, *args, **kwargs):
            super().__init__('skew5', *args, **kwargs)

    out = {}
    a = img((45, 57), 191) + np.linspace(0, 3, 57, dtype=np.float32)[None, :]
    sig = img((180,), 192) + 2 * np.sin(np.arange(180, dtype=np.float32) / 7.)
    cube = img((8, 14, 16), 193) + np.linspace(0, 2, 16, dtype=np.float32)[None, None, :]
    var = (0.5 + np.abs(img((45, 57), 194))).astype(np.float32)
    out["img"], out["sig"], out["cube"], out["var"] = a.astype(np.float32), sig.astype(np.float32), \
        cube.astype(np.float32), var
    a, sig, cube = out["img"], out["sig"], out["cube"]
    for name, cls in (("bin7", Binomial7), ("skew5", Skew5)):
        out[f"{name}_taps"] = cls.coefficients_1d
        out[f"{name}_b2d_L3"] = AtrousTransform(cls, bilateral=1)(a, 3).data
        out[f"{name}_b2d_list_L2"] = AtrousTransform(cls, bilateral=[1.5, .7],
                                                     bilateral_scaling=True)(a, 2).data
        out[f"{name}_b1d_L3"] = AtrousTransform(cls, bilateral=1)(sig, 3).data
        out[f"{name}_c3d_L2"] = AtrousTransform(cls)(cube, 2).data
        out[f"{name}_b3d_L2"] = AtrousTransform(cls, bilateral=1)(cube, 2).data
        out[f"{name}_conv3d_s1"] = convolution(cube, cls(3), s=1)
        out[f"{name}_sdev_s1"] = sdev_loc(a, cls(2), s=1)
        out[f"{name}_var_s0"] = sdev_loc(a, cls(2), s=0, variance=True)
        k2 = cls(2).kernel.astype(np.float32)
        out[f"{name}_ac_var_s1"] = atrous_convolution(a, k2, var, s=1)
        out[f"{name}_ac_plain_s2"] = atrous_convolution(a, k2, None, s=2)
        out[f"{name}_rec2_b1_L2"] = AtrousTransform(cls, bilateral=1)(a, 2, recursive=True).data
        out[f"{name}_rec1_b1_L2"] = AtrousTransform(cls, bilateral=1)(sig, 2, recursive=True).data
        out[f"{name}_rec3_L2"] = AtrousTransform(cls)(cube, 2, recursive=True).data
    save("g19_custom_bilateral_nd", "bilateral taps: hard (numpy); smoothing: semantic(cv2 stand-in), 1-D scipy", **out)


def g20_float64():
    """float64 and integer inputs: the reference computes them in float64 (wavelets.py:297,
    319-320).  Pins the float64 engine at double-precision tolerance."""
    from watroo.wavelets import AbstractScalingFunction, sdev_loc
    from watroo.utils import enhance

    class Skew5(AbstractScalingFunction):
        coefficients_1d = np.array([0.05, 0.25, 0.4, 0.2, 0.1])
        sigma_e_1d = np.array([0.7, 0.3, 0.2, 0.12, 0.08, 0.06])
        sigma_e_2d = np.array([0.9, 0.2, 0.09, 0.04, 0.02, 0.01])

        def __init__(self, *args, **kwargs):
            super().__init__('skew5', *args, **kwargs)

    rng = np.random.default_rng(201)
    out = {}
    a = rng.standard_normal((45, 57)) * 1e3 + 1e5 + 50 * np.sin(np.arange(57) / 5.)[None, :]   # large offset: float32 would lose it
    sig = rng.standard_normal(200) + 1e4
    cube = rng.standard_normal((8, 12, 14))
    pos = np.abs(rng.standard_normal((30, 34))) * 20 + 5
    ints = rng.integers(0, 60000, (40, 36)).astype(np.int32)
    u16 = rng.integers(0, 65535, (33, 29)).astype(np.uint16)
    out.update(img=a, sig=sig, cube=cube, pos=pos, ints=ints, u16=u16)
    assert a.dtype == np.float64
    for fam, cls in FAM.items():
        out[f"{fam}_coef2_L3"] = AtrousTransform(cls)(a, 3).data
        out[f"{fam}_coef2_L5"] = AtrousTransform(cls)(a, 5).data
        out[f"{fam}_coef1_L3"] = AtrousTransform(cls)(sig, 3).data
        out[f"{fam}_coef3_L2"] = AtrousTransform(cls)(cube, 2).data
        out[f"{fam}_ints_L3"] = AtrousTransform(cls)(ints, 3).data
        out[f"{fam}_conv2_s2"] = convolution(a, cls(2), s=2)
        out[f"{fam}_conv1_s1"] = convolution(sig, cls(1), s=1)
        out[f"{fam}_conv3_s1"] = convolution(cube, cls(3), s=1)
        out[f"{fam}_sdev_s1"] = sdev_loc(a, cls(2), s=1)
        out[f"{fam}_var_s0"] = sdev_loc(a, cls(2), s=0, variance=True)
        out[f"{fam}_den2"] = denoise(a.copy(), [5, 3], cls)
        out[f"{fam}_den2_hard"] = denoise(a.copy(), [3, 2, 1], cls, soft_threshold=False)
        out[f"{fam}_den1"] = denoise(sig.copy(), [4, 2], cls)
        out[f"{fam}_den3"] = denoise(cube.copy(), [4, 2], cls)
    out["u16_coef_L2"] = AtrousTransform(B3spline)(u16, 2).data
    c = AtrousTransform(B3spline)(a, 4)
    assert c.data.dtype == np.float64
    out["noise"] = np.float64(c.get_noise())
    out["sig_soft_s1"] = c.significance(3.0, 1)
    out["sig_hard_s0"] = c.significance(2.0, 0, soft_threshold=False)
    c.denoise([5, 3, 2], weights=[1, .5, 2])
    out["den_planes"] = c.data
    cm = AtrousTransform(Triangle)(a, 3)
    cm.noise = (1 + np.abs(rng.standard_normal(a.shape))) * 900.0            # per-pixel noise map
    out["noise_map"] = cm.noise
    cm.denoise([3, 2])
    out["den_planes_noise_map"] = cm.data
    out["den_pos_anscombe"] = denoise(pos.copy(), [4, 2], anscombe=True)
    out["ans_pos"] = generalized_anscombe(pos)
    out["ans_pos_inv"] = generalized_anscombe(out["ans_pos"], inverse=True)
    out["enh"] = enhance(a.copy(), weights=[.5, 2, 1], denoise=[4, 2])
    out["skew5_coef2_L2"] = AtrousTransform(Skew5)(a, 2).data
    out["skew5_coef1_L2"] = AtrousTransform(Skew5)(sig, 2).data
    out["skew5_den2"] = denoise(a.copy(), [4, 2], Skew5)
    # wow without bilateral filtering in float64 (utils.py:105-219): image (own n_scales), keyword
    # combinations, a signal and a cube
    b = a - 1e5                                          # structure at unit scale on a small offset
    out["wow_img"] = b
    cases = {"default": dict(), "den": dict(denoise_coefficients=[5, 2], n_scales=3),
             "gamma": dict(denoise_coefficients=[4, 2], n_scales=3, h=0.5, gamma=2.5),
             "pv": dict(preserve_variance=True, weights=[0.5, 2], n_scales=3),
             "tri_hard": dict(scaling_function=Triangle, denoise_coefficients=[3, 1], soft_threshold=False, n_scales=4)}
    for name, kw in cases.items():
        r, cc = wow(b.copy(), **kw)
        assert r.dtype == np.float64 and cc.data.dtype == np.float64
        out[f"wow_{name}"], out[f"wow_{name}_coef"] = r, cc.data
    r, cc = wow(sig.copy() - 1e4, denoise_coefficients=[4, 2], n_scales=3)
    out["wow_sig"], out["wow_sig_coef"] = r, cc.data
    r, cc = wow(cube.copy(), denoise_coefficients=[4], n_scales=2)
    out["wow_cube"], out["wow_cube_coef"] = r, cc.data
    # bilateral filtering and the recursive algorithm in float64 (small arrays: K^2 / K^3 taps)
    sb_img = b[:30, :38].copy()
    sb_sig = (sig[:120] - 1e4) + 2 * np.sin(np.arange(120) / 9.)
    sb_cube = cube[:6, :10, :12] + np.linspace(0, 3, 12)[None, None, :]
    out.update(bil_img=sb_img, bil_sig=sb_sig, bil_cube=sb_cube)
    for fam, cls in FAM.items():
        out[f"{fam}_bil2_L3"] = AtrousTransform(cls, bilateral=1)(sb_img, 3).data
        out[f"{fam}_bil2_list_L2"] = AtrousTransform(cls, bilateral=[2.0, .7], bilateral_scaling=True)(sb_img, 2).data
        out[f"{fam}_bil1_L3"] = AtrousTransform(cls, bilateral=1)(sb_sig, 3).data
        out[f"{fam}_bil3_L2"] = AtrousTransform(cls, bilateral=1)(sb_cube, 2).data
        out[f"{fam}_rec2_L3"] = AtrousTransform(cls)(sb_img, 3, recursive=True).data
        out[f"{fam}_rec2_bil_L2"] = AtrousTransform(cls, bilateral=1)(sb_img, 2, recursive=True).data
        out[f"{fam}_rec1_L3"] = AtrousTransform(cls)(sb_sig, 3, recursive=True).data
        out[f"{fam}_rec3_L2"] = AtrousTransform(cls)(sb_cube, 2, recursive=True).data
    r, cc = wow(sb_img.copy(), bilateral=1, denoise_coefficients=[5, 2], n_scales=3)
    out["wow_bil"], out["wow_bil_coef"] = r, cc.data
    out["den_bil"] = denoise(sb_img.copy(), [4, 2], bilateral=1)
    # richardson_lucy and the stand-alone atrous_convolution on float64 data
    from watroo.wavelets import atrous_convolution
    yy, xx = np.mgrid[:48, :40]
    truth = 5 + 40 * np.exp(-((yy - 20) ** 2 + (xx - 15) ** 2) / 18.) + 25 * np.exp(-((yy - 33) ** 2 + (xx - 28) ** 2) / 8.)
    gk = np.exp(-(np.arange(-2, 3) ** 2) / 2.0)
    psf = np.outer(gk, gk * np.array([1, 1, 1, .8, .6]))
    psf = psf / psf.sum()
    rl_data = ndimage.correlate(truth, psf[::-1, ::-1], mode="reflect") + rng.standard_normal(truth.shape) * .5
    out["rl_data"], out["rl_psf"] = rl_data, psf
    assert rl_data.dtype == np.float64
    for name, kw in {"soft": dict(iterations=3), "hard": dict(iterations=3, threshold_type='hard'),
                     "nonpers": dict(iterations=2, persistent_mrs=False, denoise_coefficients=(4, 2)),
                     "fft": dict(iterations=3, fft=True)}.items():
        out[f"rl_{name}"] = richardson_lucy(rl_data.copy(), psf, **kw)
    k2 = B3spline(2).kernel
    vmap = 0.5 + np.abs(rng.standard_normal(sb_img.shape))
    out["ac_var"] = vmap
    out["ac_plain_s1"] = atrous_convolution(sb_img, k2, None, s=1)
    out["ac_var_s1"] = atrous_convolution(sb_img, k2, vmap, s=1)
    for k, v in out.items():
        if k not in ("ints", "u16", "sig_hard_s0", "noise"):
            assert np.asarray(v).dtype == np.float64, (k, np.asarray(v).dtype)
    save("g20_float64", "1-D: hard (numpy + scipy); 2-D / 3-D: semantic(cv2 stand-in), float64", **out)


def g14_wow_denoise_nd():
    """wow / denoise on 1-D signals and (Z, Y, X) cubes (the reference is ndim-generic)."""
    out = {}
    sig = img((300,), 61)
    cube = img((20, 24, 28), 62)
    pos = img((20, 24, 28), 63, positive=True)
    out["sig"], out["cube"], out["pos"] = sig, cube, pos
    cases = {
        "default": dict(),
        "den": dict(denoise_coefficients=[5, 2], n_scales=3),
        "gamma": dict(denoise_coefficients=[4, 2], n_scales=2, h=0.5, gamma=2.5),
        "pv": dict(preserve_variance=True, weights=[0.5, 2]),
        "tri": dict(scaling_function=Triangle, denoise_coefficients=[3]),
    }
    for tag, arr in (("sig", sig), ("cube", cube)):
        for name, kw in cases.items():
            r, c = wow(arr.copy(), **kw)
            out[f"wow_{tag}_{name}"] = r
            if name in ("den", "default"):
                out[f"wow_{tag}_{name}_coef"] = c.data
        out[f"den_{tag}"] = denoise(arr.copy(), [5, 3])
        out[f"den_{tag}_tri_hard"] = denoise(arr.copy(), [4, 2, 1], Triangle, soft_threshold=False)
        out[f"den_{tag}_noise"] = denoise(arr.copy(), [5, 3], noise=0.7)
    out["den_pos_anscombe"] = denoise(pos.copy(), [5, 3], anscombe=True)
    out["ans_pos"] = generalized_anscombe(pos.copy())
    save("g14_wow_denoise_nd", "1-D: hard; 3-D: semantic(cv2 stand-in)", **out)


def g15_custom_scaling_function():
    """User-defined scaling functions: AbstractScalingFunction subclasses with their own taps
    (symmetric 7-tap binomial; an ASYMMETRIC 5-tap one, which pins filter2D's correlation
    orientation in 2-D and ndimage.convolve's convolution orientation in 1-D)."""
    from watroo.wavelets import AbstractScalingFunction

    class Binomial7(AbstractScalingFunction):
        coefficients_1d = np.array([1, 6, 15, 20, 15, 6, 1]) / 64
        sigma_e_1d = np.array([0.8, 0.25, 0.15, 0.1, 0.07, 0.05])
        sigma_e_2d = np.array([0.93, 0.17, 0.07, 0.03, 0.015, 0.0075])

        def __init__(self, *args, **kwargs):
            super().__init__('binomial7', *args, **kwargs)

    class Skew5(AbstractScalingFunction):
        coefficients_1d = np.array([0.05, 0.25, 0.4, 0.2, 0.1])
        sigma_e_1d = np.array([0.7, 0.3, 0.2, 0.12, 0.08, 0.06])
        sigma_e_2d = np.array([0.9, 0.2, 0.09, 0.04, 0.02, 0.01])

        def __init__(self, *args, **kwargs):
            super().__init__('skew5', *args, **kwargs)

    out = {}
    a = img((61, 83), 71)
    sig = img((200,), 72)
    out["img"], out["sig"] = a, sig
    for name, cls in (("bin7", Binomial7), ("skew5", Skew5)):
        out[f"{name}_taps"] = cls.coefficients_1d
        out[f"{name}_sigma_e_1d"], out[f"{name}_sigma_e_2d"] = cls.sigma_e_1d, cls.sigma_e_2d
        out[f"{name}_coef_2d_L3"] = AtrousTransform(cls)(a, 3).data
        out[f"{name}_coef_2d_L5"] = AtrousTransform(cls)(a, 5).data      # multi-bounce reflection
        out[f"{name}_coef_1d_L3"] = AtrousTransform(cls)(sig, 3).data
        out[f"{name}_conv_2d_s2"] = convolution(a, cls(2), s=2)
        out[f"{name}_conv_1d_s1"] = convolution(sig, cls(1), s=1)
        out[f"{name}_den_2d"] = denoise(a.copy(), [5, 3], cls)
        out[f"{name}_den_1d"] = denoise(sig.copy(), [4, 2], cls, noise=0.8)
        r, c = wow(a.copy(), cls, denoise_coefficients=[5, 2], n_scales=3)
        out[f"{name}_wow"], out[f"{name}_wow_coef"] = r, c.data
        out[f"{name}_rec_L3"] = AtrousTransform(cls)(a, 3, recursive=True).data
    save("g15_custom", "1-D: hard; 2-D: semantic(cv2 stand-in)", **out)


def g16_bilateral_nd():
    """Bilateral transforms of 1-D signals and 3-D cubes (atrous_convolution is ndim-generic)."""
    out = {}
    sig = img((300,), 81) + 3 * np.sin(np.arange(300, dtype=np.float32) / 9.)
    cube = img((10, 18, 22), 82) + np.linspace(0, 4, 22, dtype=np.float32)[None, None, :]
    out["sig"], out["cube"] = sig.astype(np.float32), cube.astype(np.float32)
    for tag, arr in (("sig", out["sig"]), ("cube", out["cube"])):
        for fam, cls in FAM.items():
            out[f"{tag}_{fam}_b1_L3"] = AtrousTransform(cls, bilateral=1)(arr, 3).data
            out[f"{tag}_{fam}_blist_scaling_L2"] = AtrousTransform(
                cls, bilateral=[2.0, 0.7], bilateral_scaling=True)(arr, 2).data
    c = AtrousTransform(B3spline, bilateral=1)(out["cube"], 2)
    out["cube_bilateral_noise"] = np.float64(c.get_noise())
    c.denoise([4, 2])
    out["cube_bilateral_den"] = c.data
    r, cc = wow(out["cube"].copy(), bilateral=1, n_scales=2, denoise_coefficients=[4, 2])
    out["cube_wow_bilateral"] = r
    save("g16_bilateral_nd", "1-D: hard (numpy + scipy mirror); 3-D: semantic(cv2 stand-in)", **out)


def g10_enhance():
    """utils.enhance (importable by path, not in __all__)."""
    from watroo.utils import enhance
    out = {}
    a = img((40, 36), 31)
    rgb = np.stack([img((40, 36), 32 + i) for i in range(3)])
    out["img"], out["rgb"] = a, rgb
    out["enh_2d"] = enhance(a.copy(), weights=[.5, 2, 1], denoise=[4, 2])
    out["enh_2d_noise"] = enhance(a.copy(), 0.8, weights=[1.5], denoise=[3, 2], soft_threshold=False)
    out["enh_rgb"] = enhance(rgb.copy(), weights=[[.5, 2], [1], [2, 2, 1]], denoise=[[3], [4, 2], None])
    out["enh_rgb_tri"] = enhance(rgb.copy(), weights=2., denoise=3., scaling_function_class=Triangle)
    save("g10_enhance", "semantic(cv2 stand-in)", **out)


def g11_one_dimensional():
    """1-D branch (scipy.ndimage.convolve, mode='mirror'): no cv2 involved -> hard pin."""
    import cv2 as _cv2
    _real = _cv2.filter2D

    def _forbidden(*a, **k):
        raise AssertionError("1-D group must not call cv2.filter2D")
    _cv2.filter2D = _forbidden
    try:
        out = {}
        for n, seed in ((300, 41), (17, 42), (5, 43)):
            a = img((n,), seed)
            out[f"sig_{n}"] = a
            for fam, cls in FAM.items():
                for L in (1, 3, 5):
                    out[f"coef_{fam}_{n}_L{L}"] = AtrousTransform(cls)(a, L).data
                out[f"conv_{fam}_{n}_s2"] = convolution(a, cls(1), s=2)
        a = img((300,), 41)
        c = AtrousTransform(B3spline)(a, 4)
        out["noise_300"] = np.float64(c.get_noise())
        c.denoise([5, 3])
        out["den_300"] = c.data
        save("g11_1d", "hard", **out)
    finally:
        _cv2.filter2D = _real


def g12_three_dimensional():
    """3-D branch (per-slice filter2D + axis-0 filter2D), SURVEY 8f rank 2."""
    out = {}
    for shape, seed in (((12, 10, 14), 51), ((5, 33, 20), 52)):
        a = img(shape, seed)
        tag = "x".join(map(str, shape))
        out[f"cube_{tag}"] = a
        for fam, cls in FAM.items():
            for L in (1, 3):
                out[f"coef_{fam}_{tag}_L{L}"] = AtrousTransform(cls)(a, L).data
            out[f"conv_{fam}_{tag}_s1"] = convolution(a, cls(3), s=1)
    a = img((12, 10, 14), 51)
    c = AtrousTransform(B3spline)(a, 3)
    out["noise_3d"] = np.float64(c.get_noise())
    c.denoise([5, 3])
    out["den_3d"] = c.data
    save("g12_3d", "semantic(cv2 stand-in)", **out)



def g21_general_operator():
    """Round 3: what the HIP engine used to refuse.  (a) atrous_convolution (the reference's own numpy
    loop, wavelets.py:74-105 - no cv2: HARD pin) with non-separable, rectangular and even-sized
    kernels, every np.pad mode the engine implements, with and without the range weights, on
    signals, images and cubes; (b) user-defined scaling functions with an EVEN number of taps and
    with more than 15 taps through convolution() / AtrousTransform (cv2 stand-in: semantic pin;
    1-D: scipy itself, hard)."""
    import cv2 as _cv2
    from watroo.wavelets import AbstractScalingFunction
    out = {}
    rng = np.random.default_rng(2100)
    a = img((37, 53), 211)
    sig = img((120,), 212)
    cube = img((6, 9, 11), 213)
    var = (np.abs(img((37, 53), 214)) + 0.3).astype(np.float32)
    out["img"], out["sig"], out["cube"], out["var"] = a, sig, cube, var
    kernels = {"k3x3": rng.random((3, 3)), "k3x5": rng.random((3, 5)), "k4x4": rng.random((4, 4)),
               "k2x2": rng.random((2, 2)), "k5x1": rng.random((5, 1))}
    for name, k in kernels.items():
        kernels[name] = k / k.sum()
        out[name] = kernels[name]
    out["k1d4"] = rng.random(4)
    out["k1d4"] /= out["k1d4"].sum()
    out["k3d"] = rng.random((3, 3, 3))
    out["k3d"] /= out["k3d"].sum()
    _real = _cv2.filter2D

    def _forbidden(*args, **kw):
        raise AssertionError("hard-pin group must not call cv2.filter2D")
    _cv2.filter2D = _forbidden
    try:
        for name, k in kernels.items():
            for mode in ("symmetric", "reflect", "edge", "wrap", "constant"):
                for s in (0, 2):
                    out[f"ac_{name}_{mode}_s{s}"] = atrous_convolution(a, k, s=s, mode=mode)
            out[f"acb_{name}_symmetric_s1"] = atrous_convolution(a, k, var, s=1)
            out[f"acb_{name}_reflect_s1"] = atrous_convolution(a, k, var, s=1, mode="reflect")
        out["ac_f64_k4x4_s1"] = atrous_convolution(a.astype(np.float64) * 1e3 + 7e5, kernels["k4x4"], s=1)
        out["acb_f64_k3x3_s1"] = atrous_convolution(a.astype(np.float64), kernels["k3x3"], var.astype(np.float64), s=1,
                                                    mode="edge")
        for mode in ("symmetric", "reflect", "wrap"):
            out[f"ac1_{mode}_s1"] = atrous_convolution(sig, out["k1d4"], s=1, mode=mode)
            out[f"ac3_{mode}_s1"] = atrous_convolution(cube, out["k3d"], s=1, mode=mode)
        out["acb3_symmetric_s0"] = atrous_convolution(cube, out["k3d"], np.float32(0.7), s=0)
    finally:
        _cv2.filter2D = _real

    class Haar2(AbstractScalingFunction):
        coefficients_1d = np.array([0.5, 0.5])
        sigma_e_1d = np.array([0.7, 0.35, 0.18, 0.09, 0.045, 0.02])
        sigma_e_2d = np.array([0.87, 0.22, 0.1, 0.05, 0.025, 0.012])
        sigma_e_3d = np.array([0.95, 0.12, 0.04, 0.014, 0.005])

        def __init__(self, *args, **kwargs):
            super().__init__('haar2', *args, **kwargs)

    class Even4(AbstractScalingFunction):
        coefficients_1d = np.array([0.1, 0.4, 0.3, 0.2])
        sigma_e_1d = np.array([0.7, 0.3, 0.2, 0.12, 0.08, 0.06])
        sigma_e_2d = np.array([0.9, 0.2, 0.09, 0.04, 0.02, 0.01])
        sigma_e_3d = np.array([0.95, 0.12, 0.04, 0.014, 0.005])

        def __init__(self, *args, **kwargs):
            super().__init__('even4', *args, **kwargs)

    class Long17(AbstractScalingFunction):
        coefficients_1d = np.hanning(19)[1:-1] / np.hanning(19)[1:-1].sum()
        sigma_e_1d = np.array([0.5, 0.3, 0.2, 0.12, 0.08, 0.06])
        sigma_e_2d = np.array([0.6, 0.2, 0.09, 0.04, 0.02, 0.01])
        sigma_e_3d = np.array([0.7, 0.12, 0.04, 0.014, 0.005])

        def __init__(self, *args, **kwargs):
            super().__init__('long17', *args, **kwargs)

    for name, cls in (("haar2", Haar2), ("even4", Even4), ("long17", Long17)):
        out[f"{name}_taps"] = cls.coefficients_1d
        out[f"{name}_conv2_s1"] = convolution(a, cls(2), s=1)
        out[f"{name}_conv1_s2"] = convolution(sig, cls(1), s=2)
        out[f"{name}_coef2_L3"] = AtrousTransform(cls)(a, 3).data
        out[f"{name}_coef1_L2"] = AtrousTransform(cls)(sig, 2).data
        out[f"{name}_den2"] = denoise(a.copy(), [5, 3], cls)
    out["even4_coef3_L2"] = AtrousTransform(Even4)(cube, 2).data
    out["even4_coef2_f64_L2"] = AtrousTransform(Even4)(a.astype(np.float64) + 1e4, 2).data
    save("g21_general", "atrous_convolution: hard; scaling functions: 1-D hard, 2-D/3-D semantic(cv2 stand-in)", **out)

def g22_remaining_refusals():
    """Round 4: the last cases the HIP engine refused.  (a) atrous_convolution under the np.pad modes
    the engine does not implement on the device ('linear_ramp', 'maximum', 'mean', 'median', 'minimum':
    the reference hands `mode` to np.pad, wavelets.py:77) - the reference's own numpy loop, cv2
    forbidden: HARD pin.  (b) bilateral and recursive transforms with scaling functions of an even
    number of taps or more than 15 (wavelets.py:330-406, 433-440): signals / images / cubes (cv2
    stand-in for convolution(): semantic pin; 1-D: scipy itself)."""
    import cv2 as _cv2
    from watroo.wavelets import AbstractScalingFunction
    out = {}
    rng = np.random.default_rng(2200)
    a = img((41, 58), 221)
    sig = img((150,), 222)
    cube = img((9, 12, 10), 223)
    var = (np.abs(img((41, 58), 224)) + 0.3).astype(np.float32)
    out["img"], out["sig"], out["cube"], out["var"] = a, sig, cube, var
    k33, k42 = rng.random((3, 3)), rng.random((4, 2))
    out["k3x3"], out["k4x2"] = k33 / k33.sum(), k42 / k42.sum()
    out["k1d3"] = np.array([0.2, 0.5, 0.3])
    _real = _cv2.filter2D

    def _forbidden(*args, **kw):
        raise AssertionError("hard-pin group must not call cv2.filter2D")
    _cv2.filter2D = _forbidden
    try:
        for mode in ("linear_ramp", "maximum", "mean", "median", "minimum"):
            for name in ("k3x3", "k4x2"):
                for s in (0, 2):
                    out[f"ac_{name}_{mode}_s{s}"] = atrous_convolution(a, out[name], s=s, mode=mode)
            out[f"acb_k3x3_{mode}_s1"] = atrous_convolution(a, out["k3x3"], var, s=1, mode=mode)
            out[f"ac1_{mode}_s1"] = atrous_convolution(sig, out["k1d3"], s=1, mode=mode)
        out["ac_f64_k4x2_mean_s1"] = atrous_convolution(a.astype(np.float64) * 1e3 + 7e5, out["k4x2"], s=1, mode="mean")
    finally:
        _cv2.filter2D = _real

    class Even4(AbstractScalingFunction):
        coefficients_1d = np.array([0.1, 0.4, 0.3, 0.2])
        sigma_e_1d = np.array([0.7, 0.3, 0.2, 0.12, 0.08, 0.06])
        sigma_e_2d = np.array([0.9, 0.2, 0.09, 0.04, 0.02, 0.01])
        sigma_e_3d = np.array([0.95, 0.12, 0.04, 0.014, 0.005])

        def __init__(self, *args, **kwargs):
            super().__init__('even4', *args, **kwargs)

    class Long17(AbstractScalingFunction):
        coefficients_1d = np.hanning(19)[1:-1] / np.hanning(19)[1:-1].sum()
        sigma_e_1d = np.array([0.5, 0.3, 0.2, 0.12, 0.08, 0.06])
        sigma_e_2d = np.array([0.6, 0.2, 0.09, 0.04, 0.02, 0.01])
        sigma_e_3d = np.array([0.7, 0.12, 0.04, 0.014, 0.005])

        def __init__(self, *args, **kwargs):
            super().__init__('long17', *args, **kwargs)

    small = img((12, 14), 225)                     # (Long17's 3-D bilateral kernel has 17^3 taps: a small cube)
    out["small"] = small
    for name, cls in (("even4", Even4), ("long17", Long17)):
        out[f"{name}_taps"] = cls.coefficients_1d
        out[f"{name}_bil2_L3"] = AtrousTransform(cls, bilateral=1)(a, 3).data
        out[f"{name}_bil2_scaled_L2"] = AtrousTransform(cls, bilateral=[1.5, 0.7], bilateral_scaling=True)(a, 2).data
        out[f"{name}_bil1_L2"] = AtrousTransform(cls, bilateral=2)(sig, 2).data
        out[f"{name}_rec2_L3"] = AtrousTransform(cls)(a, 3, recursive=True).data
        out[f"{name}_rec1_L3"] = AtrousTransform(cls)(sig, 3, recursive=True).data
        out[f"{name}_recbil2_L2"] = AtrousTransform(cls, bilateral=1)(a, 2, recursive=True).data
        out[f"{name}_recbil1_L2"] = AtrousTransform(cls, bilateral=1)(sig, 2, recursive=True).data
    out["even4_bil3_L2"] = AtrousTransform(Even4, bilateral=1)(cube, 2).data
    out["even4_rec3_L2"] = AtrousTransform(Even4)(cube, 2, recursive=True).data
    out["even4_recbil3_L2"] = AtrousTransform(Even4, bilateral=1)(cube, 2, recursive=True).data
    out["even4_rec2_f64_L3"] = AtrousTransform(Even4)(a.astype(np.float64) + 1e4, 3, recursive=True).data
    out["even4_bil2_f64_L2"] = AtrousTransform(Even4, bilateral=1)(a.astype(np.float64) + 1e4, 2).data
    save("g22_refusals", "atrous_convolution pad modes: hard; transforms: 1-D hard, 2-D/3-D semantic(cv2 stand-in)", **out)


def g23_richardson_lucy_fft_large_psf():
    """Round 4: richardson_lucy(fft=True) with a LARGE PSF (25 x 23 and 24 x 32 taps) on power-of-two
    images - the shapes the HIP engine now serves through its own FFT (numpy rfft2 / irfft2 in the
    reference's loop, utils.py:245-254, 284; the a-trous transform inside through the cv2 stand-in)."""
    rng = np.random.default_rng(2300)
    yy, xx = np.mgrid[0:64, 0:128]
    truth = (np.exp(-((yy - 20.) ** 2 + (xx - 40.) ** 2) / 30.) * 40 + np.exp(-((yy - 45.) ** 2 + (xx - 90.) ** 2) / 80.) * 25
             + 2.0).astype(np.float32)
    ky, kx = np.mgrid[0:25, 0:23]
    psf = np.exp(-((ky - 12.) ** 2 / 40. + (kx - 11.) ** 2 / 25. + 0.02 * (ky - 12.) * (kx - 11.))).astype(np.float32)
    psf /= psf.sum()
    even = rng.uniform(0.2, 1.0, (24, 32)).astype(np.float32) * np.hanning(24)[:, None].astype(np.float32) \
        * np.hanning(32)[None, :].astype(np.float32)
    even /= even.sum()
    data = (truth + rng.standard_normal(truth.shape).astype(np.float32) * 0.5).astype(np.float32)
    out = {"data": data, "psf": psf, "psf_even": even}
    out["rl_fft_soft"] = richardson_lucy(data.copy(), psf, iterations=4, fft=True)
    out["rl_fft_hard"] = richardson_lucy(data.copy(), psf, iterations=3, threshold_type='hard', fft=True, persistent_mrs=False)
    out["rl_fft_even"] = richardson_lucy(data.copy(), even, iterations=3, fft=True, denoise_coefficients=(4, 2))
    out["rl_fft_f64"] = richardson_lucy(data.astype(np.float64) * 10 + 100, psf.astype(np.float64), iterations=3, fft=True)
    for name, k in (("psf", psf), ("psf_even", even)):
        pad = np.zeros_like(data)
        H, W = data.shape
        kh, kw = k.shape
        pad[H // 2 - kh // 2:H // 2 - kh // 2 + kh, W // 2 - kw // 2:W // 2 - kw // 2 + kw] = k
        f = np.fft.rfft2(np.roll(pad, (H // 2, W // 2), axis=(0, 1)))
        out[f"circ_conv_{name}"] = np.fft.irfft2(np.fft.rfft2(data) * f)
        out[f"circ_corr_{name}"] = np.fft.irfft2(np.fft.rfft2(data) * f.conj())
    save("g23_rl_fft_large", "hard(numpy fft; transform via cv2 stand-in)", **out)


def g24_richardson_lucy_fft_non_pow2():
    """Round 4: richardson_lucy(fft=True) with a large PSF on images whose sides are NOT powers of two
    (72 x 100, and 75 x 100: an odd height moves the row anchors, utils.py:246-250) - the HIP engine
    serves them through a periodically extended power-of-two FFT."""
    rng = np.random.default_rng(2400)
    ky, kx = np.mgrid[0:25, 0:23]
    psf = np.exp(-((ky - 12.) ** 2 / 40. + (kx - 11.) ** 2 / 25. + 0.02 * (ky - 12.) * (kx - 11.))).astype(np.float32)
    psf /= psf.sum()
    even = rng.uniform(0.2, 1.0, (24, 26)).astype(np.float32) * np.hanning(24)[:, None].astype(np.float32) \
        * np.hanning(26)[None, :].astype(np.float32)
    even /= even.sum()
    out = {"psf": psf, "psf_even": even}
    for tag, (H, W) in (("a", (72, 100)), ("odd", (75, 100))):
        yy, xx = np.mgrid[0:H, 0:W]
        truth = (np.exp(-((yy - 20.) ** 2 + (xx - 40.) ** 2) / 30.) * 40 + np.exp(-((yy - 50.) ** 2 + (xx - 70.) ** 2) / 80.) * 25
                 + 2.0).astype(np.float32)
        data = (truth + rng.standard_normal(truth.shape).astype(np.float32) * 0.5).astype(np.float32)
        out[f"data_{tag}"] = data
        out[f"rl_{tag}_soft"] = richardson_lucy(data.copy(), psf, iterations=3, fft=True)
        out[f"rl_{tag}_even"] = richardson_lucy(data.copy(), even, iterations=3, fft=True, denoise_coefficients=(4, 2))
        out[f"rl_{tag}_f64"] = richardson_lucy(data.astype(np.float64) * 10 + 100, psf.astype(np.float64), iterations=3, fft=True)
    save("g24_rl_fft_nonpow2", "hard(numpy fft; transform via cv2 stand-in)", **out)


if __name__ == "__main__":
    if "--only" in sys.argv:                      # e.g. --only g13_richardson_lucy_fft
        globals()[sys.argv[sys.argv.index("--only") + 1]]()
    elif REAL_NE:
        assert NE_KIND.startswith("real"), "run with /opt/conda/bin/python3.9"
        g5_bilateral("g5_realne")
    else:
        g0_hard()
        g1_transform()
        g2_g3_denoise()
        g4_wow()
        g5_bilateral()
        g7_recursive_g8_tests()
        g9_richardson_lucy()
        g13_richardson_lucy_fft()
        g10_enhance()
        g11_one_dimensional()
        g12_three_dimensional()
        g14_wow_denoise_nd()
        g15_custom_scaling_function()
        g16_bilateral_nd()
        g17_rl_fft_odd_height()
        g18_recursive_nd_bilateral()
        g19_custom_bilateral_nd()
        g20_float64()
        g21_general_operator()
        g22_remaining_refusals()
        g23_richardson_lucy_fft_large_psf()
        g24_richardson_lucy_fft_non_pow2()
