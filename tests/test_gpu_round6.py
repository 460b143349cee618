"""Round-6 GPU tests: fast addressing of the fused passes at any width (float64 side), sequences of frames through the
double-buffered batch entry points, replicas."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as entry
    entry.build()
    from wavelets_amd import _lib
    if _lib.device_count() == 0:
        pytest.skip("no GPU")
    return _lib


@pytest.mark.parametrize("H,W,fam,level", [(130, 261, "b3spline", 6), (301, 1001, "b3spline", 6), (700, 1101, "b3spline", 6),
                                           (90, 129, "triangle", 4), (600, 533, "triangle", 8), (64, 51, "b3spline", 2)])
def test_float64_fused_fast_addressing_at_odd_widths_is_bitwise_the_generic_one(L, H, W, fam, level):
    """wt_fused_kernel<double>: a lane owns two pixels, so an odd width leaves one pair that straddles the right border
    ((g0, g0) after the swizzle) and reversed pairs beyond it; the fast addressing must write the bits of the generic
    (gather) addressing - planes, carried sum, and the plain decomposition."""
    import wavelets_amd as WA
    cls = {"b3spline": WA.B3spline, "triangle": WA.Triangle}[fam]
    a = np.random.default_rng(H * 7 + W).standard_normal((H, W)) * 3 + 1e3
    plan = L.Plan64(L.default_context(), H, W, tuple(float(t) for t in cls.coefficients_1d), level)
    plan.upload(L.PLANE_INPUT, a)
    got = {}
    try:
        for mode in (1, 0):
            L.set_option("fused_fast", mode)
            assert plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT)
            out = [plan.download(s).copy() for s in range(level + 1)] + [plan.download(L.PLANE_OUT).copy()]
            plan.decompose(L.PLANE_INPUT, level)
            got[mode] = out + [plan.download(s).copy() for s in range(level + 1)]
    finally:
        L.set_option("fused_fast", 1)
    for i, (x, y) in enumerate(zip(got[1], got[0])):
        np.testing.assert_array_equal(x.view(np.uint64), y.view(np.uint64), err_msg=f"output {i}")
    plan.close()
