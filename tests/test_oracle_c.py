"""The C restatement must be bit-identical to the pinned numpy oracle.  CPU only."""
import numpy as np
import pytest

from oracle import atrous_numpy as O
from oracle import cref

FAMS = ("b3spline", "triangle")


@pytest.mark.parametrize("shape", [(37, 53), (64, 48), (16, 16), (5, 3), (200, 301)])
def test_decompose_bitwise(shape):
    a = np.random.default_rng(7).standard_normal(shape).astype(np.float32)
    for fam in FAMS:
        for L in (1, 3, 5):
            np.testing.assert_array_equal(cref.decompose(a, L, fam), O.atrous_standard(a, L, fam))


def test_smooth_square_sum_median_denoise():
    a = np.random.default_rng(8).standard_normal((61, 77)).astype(np.float32)
    for fam in FAMS:
        for s in (0, 2, 4):
            np.testing.assert_array_equal(cref.smooth(a, fam, s, True),
                                          O.convolution(a ** 2, fam, s))
    stack = O.atrous_standard(a, 4)
    np.testing.assert_array_equal(cref.plane_sum(stack), stack.sum(axis=0))
    for n in (a.size, a.size - 1):
        v = a.ravel()[:n]
        assert cref.abs_median(v) == np.median(np.abs(v))
    c = O.Coeffs(stack.copy(), "b3spline")
    c.denoise([5, 3], weights=[.5, 2])
    noise = c.noise
    got = stack.copy()
    for s, (sig, w) in enumerate(zip([5, 3], [.5, 2])):
        cref.denoise_plane(got[s], sig * noise * c.sigma_e[s], w, True)
    np.testing.assert_array_equal(got, c.data)
    c = O.Coeffs(stack.copy(), "b3spline")
    c.denoise([5, 3], soft_threshold=False)
    got = stack.copy()
    for s, sig in enumerate([5, 3]):
        cref.denoise_plane(got[s], sig * noise * c.sigma_e[s], 1, False)
    np.testing.assert_array_equal(got, c.data)


def _img(shape, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) + 3 * np.sin(np.arange(shape[1]) / 7.)[None, :]).astype(np.float32)


@pytest.mark.parametrize("fam", FAMS)
def test_variance_bitwise_and_bilateral_close(fam):
    """orc_variance is IEEE-exact like the rest; orc_bilateral differs from the numpy oracle only
    through expf (libm) vs numpy's float exp: <= 2e-6 * max|a|, ten times inside the parity
    tolerance of the bilateral GPU tests (2e-5 * max|a|)."""
    a = _img((61, 77), 3)
    amax = np.abs(a).max()
    for s in (0, 1, 3, 5):
        np.testing.assert_array_equal(cref.variance(a, fam, s), O.sdev_loc(a, fam, s, variance=True))
        np.testing.assert_array_equal(cref.variance(a, fam, s, 2.25, 3.0),
                                      O.sdev_loc(a, fam, s, variance=True) * 1.5 ** 2 * 3)
        var = O.sdev_loc(a, fam, s, variance=True)
        ref = O.atrous_convolution(a, O.kernel_2d(fam, a.dtype), var, s)
        np.testing.assert_allclose(cref.bilateral(a, var, fam, s), ref, rtol=0, atol=2e-6 * amax)
    for bil, scaling in ((1, False), ([1.5, 0.7], True)):
        ref = O.atrous_standard(a, 4, fam, bil, scaling)
        np.testing.assert_allclose(cref.decompose_bilateral(a, 4, fam, bil, scaling), ref,
                                   rtol=0, atol=4e-6 * amax)


WOW_C_CASES = [
    dict(),
    dict(denoise_coefficients=[5, 2]),
    dict(n_scales=3, weights=[.5], denoise_coefficients=[5, 2]),
    dict(h=.5, gamma=2, denoise_coefficients=[5, 2]),
    dict(h=1, denoise_coefficients=[5, 2]),
    dict(preserve_variance=True, denoise_coefficients=[4]),
    dict(whitening=False, weights=[2, .5, 3]),
    dict(soft_threshold=False, denoise_coefficients=[3, 3, 1], noise=0.8),
]


@pytest.mark.parametrize("kw", WOW_C_CASES)
def test_wow_without_bilateral_is_bitwise_the_numpy_oracle(kw):
    """no exp on this path: the C-backed wow must reproduce atrous_numpy.wow bit for bit
    (scipy's erf and libm's erf agree on these inputs after the rounding to float32)"""
    a = _img((64, 96), 5)
    for fam in FAMS:
        ref_img, ref_c = O.wow(a.copy(), fam, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        img, planes = cref.wow(a.copy(), fam, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        np.testing.assert_array_equal(planes, ref_c.data)
        np.testing.assert_array_equal(img, ref_img)


@pytest.mark.parametrize("kw", [dict(bilateral=1), dict(bilateral=1, denoise_coefficients=[5, 2]),
                                dict(bilateral=[1.5, 1], bilateral_scaling=True, h=.3,
                                     denoise_coefficients=[5])])
def test_wow_bilateral_close_to_the_numpy_oracle(kw):
    a = _img((64, 96), 6)
    ref_img, ref_c = O.wow(a.copy(), "b3spline", **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    img, planes = cref.wow(a.copy(), "b3spline", **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    np.testing.assert_allclose(planes, ref_c.data, rtol=2e-5, atol=2e-5 * np.abs(ref_c.data).max())
    np.testing.assert_allclose(img, ref_img, rtol=2e-5, atol=2e-5 * np.abs(ref_img).max())
