"""Pin the numpy oracle (oracle/atrous_numpy.py) against fixtures generated from the
unmodified reference (tests/golden/make_golden.py).  CPU only.

Tolerances (fp32, stated per SURVEY.md section 7 "Tolerance statement"):
  coefficients      atol = 1e-5 * max|input|   (sum-order / FMA / direct-vs-DFT rounding)
  hard pins         the oracle follows the reference's numpy op order -> exact or 1 ulp
"""
import numpy as np
import pytest

from oracle import atrous_numpy as O
from conftest import load_golden

FAMS = ("b3spline", "triangle")
SHAPES = ("37x53", "64x48", "16x16")


def close(a, b, atol, rtol=0.0):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64),
                               rtol=rtol, atol=atol)


# ---------------------------------------------------------------- hard pins (no cv2 involved)
def test_hard_atrous_convolution_exact():
    g = load_golden("g0_hard")
    for tag in SHAPES:
        a = g[f"img_{tag}"]
        for fam in FAMS:
            for s in range(5):
                got = O.convolution(a, fam, s)
                ref = g[f"aconv_{fam}_{tag}_s{s}"]
                assert got.dtype == ref.dtype == np.float32
                np.testing.assert_array_equal(got, ref)     # same op order -> bit exact


def test_hard_coefficients_methods():
    g = load_golden("g0_hard")
    stack = g["stack"]
    for fam in FAMS:
        c = O.Coeffs(stack.copy(), fam)
        assert c.get_noise() == g[f"noise_{fam}"]
        np.testing.assert_array_equal(c.significance(3, 2, soft_threshold=False),
                                      g[f"sig_hard_3_2_{fam}"])
        np.testing.assert_array_equal(c.significance(3, 1), g[f"sig_soft_3_1_{fam}"])
        for key, sig, kw, pre in [
            ("den_53", [5, 3], {}, None),
            ("den_532_w", [5, 3, 2], dict(weights=[.5, 2, 1]), None),
            ("den_53_hard", [5, 3], dict(soft_threshold=False), None),
            ("den_32_noise07", [3, 2], {}, 0.7),
            ("den_32_noisemap", [3, 2], {}, g["noise_map"]),
        ]:
            c = O.Coeffs(stack.copy(), fam)
            if pre is not None:
                c.noise = pre
            c.denoise(sig, **kw)
            np.testing.assert_array_equal(c.data, g[f"{key}_{fam}"])
        c = O.Coeffs(stack.copy(), fam, bilateral=[1, 1])
        c.denoise([5, 3])
        np.testing.assert_array_equal(c.data, g[f"den_53_bilat_{fam}"])
    cz = O.Coeffs(np.zeros((3, 8, 8), np.float32), "b3spline")
    cz.denoise([5, 3])
    np.testing.assert_array_equal(cz.data, g["den_zero_noise_branch"])
    assert cz.noise == g["noise_zero"] == 0


def test_hard_anscombe():
    g = load_golden("g0_hard")
    p = g["ans_in"]
    np.testing.assert_array_equal(O.generalized_anscombe(p), g["ans_fwd"])
    np.testing.assert_array_equal(O.generalized_anscombe(p, alpha=2., g=1., sigma=.5),
                                  g["ans_fwd_params"])
    np.testing.assert_array_equal(
        O.generalized_anscombe(O.generalized_anscombe(p), inverse=True), g["ans_inv"])


# ------------------------------------------------- semantic pins (full API, cv2 stand-in)
def test_transform_vs_reference_api():
    g = load_golden("g1_transform")
    for tag in SHAPES:
        a = g[f"img_{tag}"]
        tol = 1e-5 * np.abs(a).max()
        for fam in FAMS:
            for L in (1, 2, 3, 4, 5):
                got = O.atrous_standard(a, L, fam)
                ref = g[f"coef_{fam}_{tag}_L{L}"]
                assert got.shape == ref.shape == (L + 1,) + a.shape and got.dtype == ref.dtype
                close(got, ref, tol)
            close(O.convolution(a, fam, 2), g[f"conv_{fam}_{tag}_s2"], tol)


def test_denoise_vs_reference_api():
    g = load_golden("g2_denoise")
    a = g["img"]
    tol = 1e-5 * np.abs(a).max()
    for fam in FAMS:
        c = O.Coeffs(O.atrous_standard(a, 4, fam), fam)
        np.testing.assert_allclose(c.get_noise(), g[f"noise_{fam}"], rtol=1e-6)
        c.denoise([5, 3])
        close(c.data, g[f"coef_den_53_{fam}"], tol)
        close(O.denoise(a, [5, 3], fam), g[f"denoise_53_{fam}"], tol)
        close(O.denoise(a, [5, 3], fam, noise=0.9), g[f"denoise_53_noise_{fam}"], tol)
        # hard threshold: a pixel within rounding of tau may flip; allow a handful
        got = O.denoise(a, [5, 3, 2], fam, soft_threshold=False)
        ref = g[f"denoise_532_hard_{fam}"]
        assert (np.abs(got - ref) > tol).sum() <= 2
    p = g["img_pos"]
    close(O.denoise(p, [5, 3], "triangle", anscombe=True), g["denoise_53_anscombe"],
          1e-5 * np.abs(p).max())


WOW_CASES = {
    "default": dict(),
    "triangle": dict(family="triangle"),
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


@pytest.mark.parametrize("name", sorted(WOW_CASES))
def test_wow_vs_reference_api(name):
    g = load_golden("g4_wow")
    a = g["img"]
    recon, coef = O.wow(a.copy(), **WOW_CASES[name])
    ref_c, ref_r = g[f"coef_{name}"], g[f"recon_{name}"]
    assert coef.data.shape == ref_c.shape          # n_scales logic (utils.py:121-138)
    # whitening divides by local power -> relative tolerance (rtol 1e-4)
    scale = np.abs(ref_c).max()
    close(coef.data, ref_c, atol=1e-4 * scale, rtol=1e-4)
    close(recon, ref_r, atol=1e-4 * max(1.0, np.abs(ref_r).max()), rtol=1e-4)
    if not np.isnan(g[f"noise_{name}"]):
        np.testing.assert_allclose(coef.noise, g[f"noise_{name}"], rtol=1e-5)


def test_wow_from_coefficients():
    g = load_golden("g4_wow")
    a = g["img"]
    c = O.Coeffs(O.atrous_standard(a.copy(), 3), "b3spline")
    recon, c2 = O.wow(c, denoise_coefficients=[5, 2])
    assert c2 is c
    close(c.data, g["coef_from_coeffs"], atol=1e-4 * np.abs(g["coef_from_coeffs"]).max(),
          rtol=1e-4)
    close(recon, g["recon_from_coeffs"], atol=1e-4 * np.abs(g["recon_from_coeffs"]).max(),
          rtol=1e-4)


@pytest.mark.parametrize("fixture", ["g5_bilateral", "g5_realne"])
def test_bilateral_vs_reference(fixture):
    """g5_realne was generated with the REAL numexpr (conda python3.9, numpy 1.26)."""
    g = load_golden(fixture)
    a = g["img"]
    tol = 2e-5 * np.abs(a).max()
    for fam in FAMS:
        k = O.kernel_2d(fam, a.dtype)
        for s in (0, 1, 2):
            var = O.sdev_loc(a, fam, s, variance=True)
            close(var, g[f"var_{fam}_s{s}"], tol)
            close(O.sdev_loc(a, fam, s), g[f"sdev_{fam}_s{s}"], tol)
            # feed the reference's variance so only the bilateral operator is compared
            close(O.atrous_convolution(a, k, g[f"var_{fam}_s{s}"], s),
                  g[f"bconv_{fam}_s{s}"], tol)
        close(O.atrous_standard(a, 3, fam, bilateral=1), g[f"coef_b1_{fam}"], 5 * tol)
        close(O.atrous_standard(a, 3, fam, bilateral=[2., .5], bilateral_scaling=True),
              g[f"coef_blist_scaling_{fam}"], 5 * tol)


def test_reference_own_tests_and_recast():
    g = load_golden("g7_misc")
    ones = np.ones((128, 128))
    got = O.atrous_standard(ones, 4)
    expected = np.zeros(got.shape)
    expected[-1] = 1                                    # tests/test_wavelets.py:8-13
    assert np.isclose(got, expected).all()
    close(got, g["ones_L4"], 1e-12)
    r, _ = O.wow(ones)                                   # tests/test_utils.py:7-9 (smoke)
    close(r, g["wow_ones"], 1e-9)
    r, _ = O.wow(ones, bilateral=True)
    close(r, g["wow_ones_bilateral"], 1e-9)
    ai = g["img_int32"]
    got = O.atrous_standard(ai, 2)
    assert got.dtype == np.float64
    close(got, g["coef_int32_L2"], 1e-9)


def test_known_answers_no_fixture():
    """impulse response reproduces sigma_e tables; perfect reconstruction."""
    imp = np.zeros((257, 257), np.float32)
    imp[128, 128] = 1
    for fam in FAMS:
        c = O.atrous_standard(imp, 5, fam)
        got = np.sqrt((c[:-1].astype(np.float64) ** 2).sum(axis=(1, 2)))
        np.testing.assert_allclose(got, O.SIGMA_E_2D[fam][:5], rtol=1e-2)  # tables are Monte-Carlo
    a = np.random.default_rng(0).standard_normal((96, 80)).astype(np.float32)
    for fam in FAMS:
        c = O.atrous_standard(a, 5, fam)
        close(c.sum(axis=0), a, 2e-6)


def test_reflect_index_matches_np_pad():
    for n in (1, 2, 5, 16):
        base = np.arange(n)
        pad = 3 * n + 2
        ref = np.pad(base, pad, mode="symmetric")
        got = O.reflect_index(np.arange(-pad, n + pad), n)
        np.testing.assert_array_equal(got, ref)


# ------------------------------------------------- richardson_lucy
RL_CASES = {
    "soft": dict(iterations=3),
    "hard": dict(iterations=3, threshold_type='hard'),
    "uniform": dict(iterations=2, uniform_init=True),
    "soft_nonpersistent": dict(iterations=3, persistent_mrs=False, denoise_coefficients=(4, 2)),
    "hard_nonpersistent": dict(iterations=2, threshold_type='hard', persistent_mrs=False),
}


@pytest.mark.parametrize("name", sorted(RL_CASES))
def test_richardson_lucy_vs_reference(name):
    g = load_golden("g9_richardson_lucy")
    got = O.richardson_lucy(g["data"].copy(), g["psf"], **RL_CASES[name])
    ref = g[f"rl_{name}"]
    close(got, ref, atol=1e-4 * np.abs(ref).max(), rtol=1e-4)


RL_FFT_CASES = {
    "rl_fft_soft": ("psf", dict(iterations=3)),
    "rl_fft_hard": ("psf", dict(iterations=2, threshold_type='hard')),
    "rl_fft_even": ("psf_even", dict(iterations=2, denoise_coefficients=(4, 2))),
}


@pytest.mark.parametrize("name", sorted(RL_FFT_CASES))
def test_richardson_lucy_fft_vs_reference(name):
    """fft=True: the oracle's direct periodic correlations against the reference's rfft2 path."""
    g = load_golden("g13_rl_fft")
    psf, kw = RL_FFT_CASES[name]
    got = O.richardson_lucy(g["data"].copy(), g[psf], fft=True, **kw)
    close(got, g[name], atol=1e-4 * np.abs(g[name]).max(), rtol=1e-4)


def test_periodic_products_even_psf_anchor():
    g = load_golden("g13_rl_fft")
    d, k = g["data"], g["psf_even"]
    kh, kw = k.shape
    tol = 1e-5 * np.abs(d).max()
    close(O.filter2d_periodic(d, k[::-1, ::-1], (kh - 1 - kh // 2, kw - 1 - kw // 2)),
          g["circ_conv_even"], tol)
    close(O.filter2d_periodic(d, k, (kh // 2, kw // 2)), g["circ_corr_even"], tol)


@pytest.mark.parametrize("tag", ["", "_thin"])
def test_periodic_products_and_rl_on_an_odd_height_image(tag):
    """g17: for an odd image height the reference's two rolls by H // 2 leave the PSF centre one
    row above the origin (utils.py:246-250): row anchors k-1-k//2-1 (forward), k//2+1 (backward),
    which may fall outside a thin PSF (np.roll-based oracle: any anchor works)."""
    g = load_golden("g17_rl_fft_odd")
    d, k = g["data"], g["psf" + tag]
    assert d.shape[0] % 2 == 1
    kh, kw = k.shape
    tol = 1e-5 * np.abs(d).max()
    close(O.filter2d_periodic(d, k[::-1, ::-1], (kh - 1 - kh // 2 - 1, kw - 1 - kw // 2)),
          g["circ_conv" + tag], tol)
    close(O.filter2d_periodic(d, k, (kh // 2 + 1, kw // 2)), g["circ_corr" + tag], tol)
    ref = g["rl_fft_odd" + tag]
    got = O.richardson_lucy(d.copy(), k, iterations=2 if tag else 3, fft=True, denoise_coefficients=(4, 2))
    close(got, ref, atol=1e-4 * np.abs(ref).max(), rtol=1e-4)


def test_filter2d_even_kernel_anchor():
    g = load_golden("g9_richardson_lucy")
    close(O.filter2d_reflect(g["data"], g["psf_even"]), g["filter_even"],
          1e-5 * np.abs(g["data"]).max())


def test_enhance_vs_reference():
    g = load_golden("g10_enhance")
    a, rgb = g["img"], g["rgb"]
    tol = 1e-5 * np.abs(a).max() * 4
    close(O.enhance(a.copy(), weights=[.5, 2, 1], denoise=[4, 2]), g["enh_2d"], tol)
    got = O.enhance(a.copy(), 0.8, weights=[1.5], denoise=[3, 2], soft_threshold=False)
    assert (np.abs(got - g["enh_2d_noise"]) > tol).sum() <= 2
    close(O.enhance(rgb.copy(), weights=[[.5, 2], [1], [2, 2, 1]], denoise=[[3], [4, 2], None]),
          g["enh_rgb"], tol)
    close(O.enhance(rgb.copy(), weights=2., denoise=3., family="triangle"), g["enh_rgb_tri"], tol)


def test_recursive_vs_reference():
    """a14: the recursive algorithm (differs from the standard one near borders at s >= 3)."""
    g = load_golden("g7_misc")
    a = g["img"]
    got = O.atrous_recursive(a, 3)
    close(got, g["recursive_b3_L3"], 1e-5 * np.abs(a).max())
    ones = np.ones((128, 128))
    assert np.isclose(O.atrous_standard(ones, 4), O.atrous_recursive(ones, 4)).all()  # ref tests/test_wavelets.py:15-19
    b = np.random.default_rng(3).standard_normal((96, 80)).astype(np.float32)
    std, rec = O.atrous_standard(b, 5), O.atrous_recursive(b, 5)
    assert np.abs(std[:3] - rec[:3]).max() < 1e-5          # identical operators up to scale 2
    assert np.abs(std[4] - rec[4]).max() > 1e-3            # but not at the borders of scale >= 3
    assert np.abs(std[4, 40:56, 30:50] - rec[4, 40:56, 30:50]).max() < 1e-5


def test_one_dimensional_hard_pin():
    """1-D branch (scipy 'mirror' border): generated without cv2 -> the oracle must match."""
    g = load_golden("g11_1d")
    for n in (300, 17, 5):
        a = g[f"sig_{n}"]
        for fam in FAMS:
            for L in (1, 3, 5):
                close(O.atrous_standard_1d(a, L, fam), g[f"coef_{fam}_{n}_L{L}"], 2e-6 * np.abs(a).max())
            close(O.convolution_1d(a, fam, 2), g[f"conv_{fam}_{n}_s2"], 2e-6 * np.abs(a).max())


def test_three_dimensional_vs_reference():
    g = load_golden("g12_3d")
    for tag in ("12x10x14", "5x33x20"):
        a = g[f"cube_{tag}"]
        tol = 1e-5 * np.abs(a).max()
        for fam in FAMS:
            for L in (1, 3):
                close(O.atrous_standard_3d(a, L, fam), g[f"coef_{fam}_{tag}_L{L}"], tol)
            close(O.convolution_3d(a, fam, 1), g[f"conv_{fam}_{tag}_s1"], tol)


# ------------------------------------------------- wow / denoise on 1-D signals and 3-D cubes
ND_WOW_CASES = {
    "default": dict(),
    "den": dict(denoise_coefficients=[5, 2], n_scales=3),
    "gamma": dict(denoise_coefficients=[4, 2], n_scales=2, h=0.5, gamma=2.5),
    "pv": dict(preserve_variance=True, weights=[0.5, 2]),
    "tri": dict(family="triangle", denoise_coefficients=[3]),
}


@pytest.mark.parametrize("tag", ["sig", "cube"])
def test_wow_and_denoise_nd_vs_reference(tag):
    g = load_golden("g14_wow_denoise_nd")
    a = g[tag]
    for name, kw in ND_WOW_CASES.items():
        kw = {k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()}
        r, c = O.wow(a.copy(), **kw)
        ref = g[f"wow_{tag}_{name}"]
        close(r, ref, atol=2e-5 * np.abs(ref).max())
        if f"wow_{tag}_{name}_coef" in g:
            close(c.data, g[f"wow_{tag}_{name}_coef"], atol=2e-5 * np.abs(ref).max())
    tol = 1e-5 * np.abs(a).max()
    close(O.denoise(a.copy(), [5, 3]), g[f"den_{tag}"], tol)
    close(O.denoise(a.copy(), [5, 3], noise=0.7), g[f"den_{tag}_noise"], tol)
    got = O.denoise(a.copy(), [4, 2, 1], "triangle", soft_threshold=False)
    assert (np.abs(got - g[f"den_{tag}_tri_hard"]) > tol).sum() <= 2
    if tag == "cube":
        close(O.denoise(g["pos"].copy(), [5, 3], anscombe=True), g["den_pos_anscombe"],
              1e-5 * np.abs(g["pos"]).max())


# ------------------------------------------------- user-defined scaling functions
@pytest.mark.parametrize("name", ["bin7", "skew5"])
def test_custom_scaling_function_vs_reference(name):
    """AbstractScalingFunction subclasses with their own taps (symmetric 7-tap, asymmetric 5-tap:
    filter2D correlates in 2-D, ndimage.convolve convolves in 1-D)."""
    g = load_golden("g15_custom")
    a, sig = g["img"], g["sig"]
    fam = O.CustomFamily(g[f"{name}_taps"], {1: g[f"{name}_sigma_e_1d"], 2: g[f"{name}_sigma_e_2d"]})
    tol = 2e-6 * np.abs(a).max()
    close(O.atrous_standard(a, 3, fam), g[f"{name}_coef_2d_L3"], tol)
    close(O.atrous_standard(a, 5, fam), g[f"{name}_coef_2d_L5"], tol)
    close(O.atrous_standard_1d(sig, 3, fam), g[f"{name}_coef_1d_L3"], tol)
    close(O.convolution(a, fam, 2), g[f"{name}_conv_2d_s2"], tol)
    close(O.convolution_1d(sig, fam, 1), g[f"{name}_conv_1d_s1"], tol)
    close(O.denoise(a.copy(), [5, 3], fam), g[f"{name}_den_2d"], 5 * tol)
    close(O.denoise(sig.copy(), [4, 2], fam, noise=0.8), g[f"{name}_den_1d"], 5 * tol)
    r, c = O.wow(a.copy(), fam, denoise_coefficients=[5, 2], n_scales=3)
    close(r, g[f"{name}_wow"], 2e-5 * np.abs(g[f"{name}_wow"]).max())
    close(O.atrous_recursive(a, 3, fam), g[f"{name}_rec_L3"], tol)


@pytest.mark.parametrize("tag", ["sig", "cube"])
def test_bilateral_nd_vs_reference(tag):
    """bilateral transforms of 1-D signals and cubes (atrous_convolution is ndim-generic)."""
    g = load_golden("g16_bilateral_nd")
    a = g[tag]
    tol = 1e-5 * np.abs(a).max()
    for fam in FAMS:
        close(O.atrous_standard_nd(a, 3, fam, 1), g[f"{tag}_{fam}_b1_L3"], tol)
        close(O.atrous_standard_nd(a, 2, fam, [2.0, 0.7], True), g[f"{tag}_{fam}_blist_scaling_L2"], tol)


def test_recursive_algorithm_nd_and_bilateral_vs_reference():
    """g18: atrous_recursive on signals / cubes and with bilateral filtering (wavelets.py:330-406);
    the oracle's generic restatement against the reference's output."""
    g = load_golden("g18_recursive_nd")
    cases = [("rec2_b1", "img2", 3, "b3spline", 1, False),
             ("rec2_blist", "img2", 3, "triangle", [1.5, .7], True),
             ("rec1_b3", "sig1", 4, "b3spline", None, False),
             ("rec1_tri", "sig1", 3, "triangle", None, False),
             ("rec1_b1", "sig1", 3, "b3spline", 1, False),
             ("rec3_tri", "cube", 2, "triangle", None, False),
             ("rec3_b3", "cube", 2, "b3spline", None, False),
             ("rec3_b1", "cube", 2, "triangle", 1, False)]
    for name, src, level, fam, bil, scaling in cases:
        got = O.atrous_recursive_nd(g[src], level, fam, bil, scaling)
        close(got, g[name], atol=3e-6 * np.abs(g[src]).max())
    # the 2-D plain case agrees with the dedicated restatement
    a = g["img2"]
    np.testing.assert_array_equal(O.atrous_recursive_nd(a, 3, "b3spline"), O.atrous_recursive(a, 3, "b3spline"))


@pytest.mark.parametrize("name", ["bin7", "skew5"])
def test_custom_taps_bilateral_nd_vs_reference(name):
    """g19: user-defined taps through the bilateral operator (2-D, 1-D, 3-D), 3-D cubes, sdev_loc,
    atrous_convolution with the class's kernel and the recursive algorithm; the asymmetric taps
    pin the orientation of every branch (filter2D correlates, scipy and the tap loop convolve)."""
    g = load_golden("g19_custom_bilateral_nd")
    a, sig, cube, var = g["img"], g["sig"], g["cube"], g["var"]
    fam = O.CustomFamily(g[f"{name}_taps"], {})
    tol = 3e-6 * np.abs(a).max()
    close(O.atrous_standard(a, 3, fam, 1), g[f"{name}_b2d_L3"], tol)
    close(O.atrous_standard(a, 2, fam, [1.5, .7], True), g[f"{name}_b2d_list_L2"], tol)
    close(O.atrous_standard_nd(sig, 3, fam, 1), g[f"{name}_b1d_L3"], tol)
    close(O.atrous_standard_nd(cube, 2, fam), g[f"{name}_c3d_L2"], tol)
    close(O.atrous_standard_nd(cube, 2, fam, 1), g[f"{name}_b3d_L2"], tol)
    close(O.convolution_3d(cube, fam, 1), g[f"{name}_conv3d_s1"], tol)
    close(O.sdev_loc(a, fam, 1), g[f"{name}_sdev_s1"], 2e-5)
    close(O.sdev_loc(a, fam, 0, variance=True), g[f"{name}_var_s0"], 2e-5)
    k2 = O.kernel_2d(fam, np.float32)
    close(O.atrous_convolution(a, k2, var, 1), g[f"{name}_ac_var_s1"], tol)
    close(O.atrous_convolution(a, k2, None, 2), g[f"{name}_ac_plain_s2"], tol)
    close(O.atrous_recursive_nd(a, 2, fam, 1), g[f"{name}_rec2_b1_L2"], tol)
    close(O.atrous_recursive_nd(sig, 2, fam, 1), g[f"{name}_rec1_b1_L2"], tol)
    close(O.atrous_recursive_nd(cube, 2, fam), g[f"{name}_rec3_L2"], tol)


def test_float64_and_integer_inputs_vs_reference():
    """g20: the reference computes float64 / integer inputs in float64 (wavelets.py:297,319-320);
    the oracle is dtype-generic and must follow to double-precision rounding."""
    g = load_golden("g20_float64")
    a, sig, cube, ints = g["img"], g["sig"], g["cube"], g["ints"]
    assert a.dtype == np.float64 and ints.dtype == np.int32
    tol = 1e-12 * np.abs(a).max()
    for fam in FAMS:
        got = O.atrous_standard(a, 3, fam)
        assert got.dtype == np.float64
        close(got, g[f"{fam}_coef2_L3"], tol)
        close(O.atrous_standard(a, 5, fam), g[f"{fam}_coef2_L5"], tol)
        close(O.atrous_standard_nd(sig, 3, fam), g[f"{fam}_coef1_L3"], 1e-12 * np.abs(sig).max())
        close(O.atrous_standard_nd(cube, 2, fam), g[f"{fam}_coef3_L2"], 1e-13)
        gi = O.atrous_standard(ints, 3, fam)
        assert gi.dtype == np.float64
        close(gi, g[f"{fam}_ints_L3"], 1e-10)
        close(O.convolution(a, fam, 2), g[f"{fam}_conv2_s2"], tol)
        close(O.denoise(a.copy(), [5, 3], fam), g[f"{fam}_den2"], tol)
        close(O.denoise(sig.copy(), [4, 2], fam), g[f"{fam}_den1"], 1e-12 * np.abs(sig).max())
    c = O.Coeffs(O.atrous_standard(a, 4, "b3spline"), "b3spline")
    assert abs(c.get_noise() - g["noise"]) <= 1e-12 * g["noise"]
    close(c.significance(3.0, 1), g["sig_soft_s1"], 1e-12)
    close(O.generalized_anscombe(g["pos"]), g["ans_pos"], 1e-12)


def test_g21_general_kernels_pad_modes_and_even_taps():
    """g21 (round 3): the oracle's restatement of the reference's atrous_convolution loop for ANY
    kernel / np.pad mode (non-separable, rectangular, even-sized; signals, images, cubes; with and
    without range weights) is BIT-identical to the reference (hard pin: pure numpy), and
    convolution_taps_nd / atrous_standard_taps_nd reproduce the reference's convolution() /
    AtrousTransform for scaling functions with an even number of taps and with 17 taps (cv2
    stand-in; 1-D: scipy itself)."""
    g = load_golden("g21_general")
    a, sig, cube, var = g["img"], g["sig"], g["cube"], g["var"]
    for name in ("k3x3", "k3x5", "k4x4", "k2x2", "k5x1"):
        k = g[name]
        for mode in ("symmetric", "reflect", "edge", "wrap", "constant"):
            for s in (0, 2):
                np.testing.assert_array_equal(O.atrous_convolution_nd(a, k, None, s, mode), g[f"ac_{name}_{mode}_s{s}"])
        for mode in ("symmetric", "reflect"):
            got = O.atrous_convolution_nd(a, k, var, 1, mode)
            np.testing.assert_allclose(got, g[f"acb_{name}_{mode}_s1"], rtol=0, atol=2e-6 * np.abs(a).max())
    np.testing.assert_array_equal(O.atrous_convolution_nd(a.astype(np.float64) * 1e3 + 7e5, g["k4x4"], None, 1),
                                  g["ac_f64_k4x4_s1"])
    for mode in ("symmetric", "reflect", "wrap"):
        np.testing.assert_array_equal(O.atrous_convolution_nd(sig, g["k1d4"], None, 1, mode), g[f"ac1_{mode}_s1"])
        np.testing.assert_array_equal(O.atrous_convolution_nd(cube, g["k3d"], None, 1, mode), g[f"ac3_{mode}_s1"])
    for name in ("haar2", "even4", "long17"):
        taps = g[f"{name}_taps"]
        tol = 2e-6 * float(np.abs(a).max())
        np.testing.assert_allclose(O.convolution_taps_nd(a, taps, 1), g[f"{name}_conv2_s1"], rtol=0, atol=tol)
        np.testing.assert_allclose(O.convolution_taps_nd(sig, taps, 2), g[f"{name}_conv1_s2"], rtol=0, atol=tol)
        np.testing.assert_allclose(O.atrous_standard_taps_nd(a, 3, taps), g[f"{name}_coef2_L3"], rtol=0, atol=tol)
        np.testing.assert_allclose(O.atrous_standard_taps_nd(sig, 2, taps), g[f"{name}_coef1_L2"], rtol=0, atol=tol)
    np.testing.assert_allclose(O.atrous_standard_taps_nd(cube, 2, g["even4_taps"]), g["even4_coef3_L2"], rtol=0,
                               atol=2e-6 * float(np.abs(cube).max()))
    np.testing.assert_allclose(O.atrous_standard_taps_nd(a.astype(np.float64) + 1e4, 2, g["even4_taps"]),
                               g["even4_coef2_f64_L2"], rtol=0, atol=1e-11 * 1e4)


def test_g22_remaining_pad_modes_bilateral_and_recursive_with_even_or_long_taps():
    """g22 (round 4): (a) atrous_convolution under np.pad's 'linear_ramp', 'maximum', 'mean', 'median',
    'minimum' - the reference's own numpy loop with cv2 forbidden: the oracle matches bit for bit;
    (b) bilateral and recursive transforms of scaling functions with 4 and 17 taps (signals, images,
    cubes, float64): the oracle's restatements atrous_standard_bilateral_taps_nd /
    atrous_recursive_taps_nd against the reference's output (1-D: scipy itself; 2-D / 3-D: cv2
    stand-in, so float rounding of the correlate differs - a few ulp)."""
    g = load_golden("g22_refusals")
    a, sig, cube, var = g["img"], g["sig"], g["cube"], g["var"]
    for mode in ("linear_ramp", "maximum", "mean", "median", "minimum"):
        for name in ("k3x3", "k4x2"):
            for s in (0, 2):
                np.testing.assert_array_equal(O.atrous_convolution_nd(a, g[name], None, s, mode), g[f"ac_{name}_{mode}_s{s}"])
        np.testing.assert_allclose(O.atrous_convolution_nd(a, g["k3x3"], var, 1, mode), g[f"acb_k3x3_{mode}_s1"], rtol=0, atol=2e-6)
        np.testing.assert_array_equal(O.atrous_convolution_nd(sig, g["k1d3"], None, 1, mode), g[f"ac1_{mode}_s1"])
    np.testing.assert_array_equal(O.atrous_convolution_nd(a.astype(np.float64) * 1e3 + 7e5, g["k4x2"], None, 1, "mean"),
                                  g["ac_f64_k4x2_mean_s1"])
    tol = 3e-6 * float(np.abs(a).max())
    for name in ("even4", "long17"):
        t = g[f"{name}_taps"]
        close(O.atrous_standard_bilateral_taps_nd(a, 3, t, 1), g[f"{name}_bil2_L3"], tol)
        close(O.atrous_standard_bilateral_taps_nd(a, 2, t, [1.5, 0.7], True), g[f"{name}_bil2_scaled_L2"], tol)
        close(O.atrous_standard_bilateral_taps_nd(sig, 2, t, 2), g[f"{name}_bil1_L2"], tol)
        close(O.atrous_recursive_taps_nd(a, 3, t), g[f"{name}_rec2_L3"], tol)
        close(O.atrous_recursive_taps_nd(sig, 3, t), g[f"{name}_rec1_L3"], tol)
        close(O.atrous_recursive_taps_nd(a, 2, t, 1), g[f"{name}_recbil2_L2"], tol)
        close(O.atrous_recursive_taps_nd(sig, 2, t, 1), g[f"{name}_recbil1_L2"], tol)
    t = g["even4_taps"]
    close(O.atrous_standard_bilateral_taps_nd(cube, 2, t, 1), g["even4_bil3_L2"], tol)
    close(O.atrous_recursive_taps_nd(cube, 2, t), g["even4_rec3_L2"], tol)
    close(O.atrous_recursive_taps_nd(cube, 2, t, 1), g["even4_recbil3_L2"], tol)
    a64 = a.astype(np.float64) + 1e4
    close(O.atrous_recursive_taps_nd(a64, 3, t), g["even4_rec2_f64_L3"], 1e-11 * 1e4)
    close(O.atrous_standard_bilateral_taps_nd(a64, 2, t, 1), g["even4_bil2_f64_L2"], 1e-10 * 1e4)


def test_g23_large_psf_fft_products_and_richardson_lucy():
    """g23 (round 4): richardson_lucy(fft=True) with 25 x 23 and 24 x 32 PSFs on a 64 x 128 image - the
    oracle's direct periodic correlations against the reference's rfft2 products (numpy only) and the
    whole iteration (float32 and float64)."""
    g = load_golden("g23_rl_fft_large")
    d = g["data"]
    tol = 2e-5 * float(np.abs(d).max())
    for name in ("psf", "psf_even"):
        k = g[name]
        kh, kw = k.shape
        close(O.filter2d_periodic(d, k[::-1, ::-1], (kh - 1 - kh // 2, kw - 1 - kw // 2)), g[f"circ_conv_{name}"], tol)
        close(O.filter2d_periodic(d, k, (kh // 2, kw // 2)), g[f"circ_corr_{name}"], tol)
    for name, psf, kw in (("rl_fft_soft", "psf", dict(iterations=4)),
                          ("rl_fft_hard", "psf", dict(iterations=3, threshold_type="hard", persistent_mrs=False)),
                          ("rl_fft_even", "psf_even", dict(iterations=3, denoise_coefficients=(4, 2)))):
        got = O.richardson_lucy(d.copy(), g[psf], fft=True, **kw)
        close(got, g[name], atol=2e-4 * np.abs(g[name]).max(), rtol=2e-4)
    got = O.richardson_lucy(d.astype(np.float64) * 10 + 100, g["psf"].astype(np.float64), iterations=3, fft=True)
    close(got, g["rl_fft_f64"], atol=1e-9 * np.abs(g["rl_fft_f64"]).max(), rtol=0)


def test_g24_richardson_lucy_fft_on_images_that_are_not_powers_of_two():
    """g24 (round 4): richardson_lucy(fft=True) with 25 x 23 and 24 x 26 PSFs on 72 x 100 and 75 x 100
    images (the odd height moves the row anchors by one, utils.py:246-250), float32 and float64."""
    g = load_golden("g24_rl_fft_nonpow2")
    for tag in ("a", "odd"):
        d = g[f"data_{tag}"]
        got = O.richardson_lucy(d.copy(), g["psf"], iterations=3, fft=True)
        close(got, g[f"rl_{tag}_soft"], atol=2e-4 * np.abs(g[f"rl_{tag}_soft"]).max(), rtol=2e-4)
        got = O.richardson_lucy(d.copy(), g["psf_even"], iterations=3, fft=True, denoise_coefficients=(4, 2))
        close(got, g[f"rl_{tag}_even"], atol=2e-4 * np.abs(g[f"rl_{tag}_even"]).max(), rtol=2e-4)
        got = O.richardson_lucy(d.astype(np.float64) * 10 + 100, g["psf"].astype(np.float64), iterations=3, fft=True)
        close(got, g[f"rl_{tag}_f64"], atol=1e-9 * np.abs(g[f"rl_{tag}_f64"]).max(), rtol=0)
