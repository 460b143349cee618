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
