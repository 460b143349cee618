"""CPU checks of the host logic behind the generic tap-list operator (wt_taps_conv): the tap lists
the Python layer builds (wavelets_amd.wavelets._reference_taps / _filter_taps) are applied here by a
numpy model of wt_taps_kernel (same border rules per axis) and must reproduce the oracle's
restatements of the reference - atrous_convolution for any kernel / np.pad mode
(watroo/wavelets.py:74-105) and convolution() for even or long tap vectors (:35-69)."""
import numpy as np
import pytest

from oracle import atrous_numpy as O
from wavelets_amd.wavelets import _PAD_MODES, _filter_taps, _reference_taps


def pad_index(i, n, mode):
    """wt_pad_index of wt_kernels_common.h"""
    i = np.asarray(i)
    if mode == "symmetric":
        m = np.mod(i, 2 * n)
        return np.where(m < n, m, 2 * n - 1 - m)
    if mode == "reflect":
        if n == 1:
            return np.zeros_like(i)
        m = np.mod(i, 2 * n - 2)
        return np.where(m < n, m, 2 * n - 2 - m)
    if mode == "edge":
        return np.clip(i, 0, n - 1)
    if mode == "wrap":
        return np.mod(i, n)
    return np.where((i >= 0) & (i < n), i, -1)              # constant


def apply_taps(a, center, offs, wts, mode, fill=0.0):
    """numpy model of wt_taps_kernel (plain form) on a 1-, 2- or 3-D array"""
    a3 = a.reshape((1,) * (3 - a.ndim) + a.shape)
    Z, Y, X = a3.shape
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    acc = (center * a3).astype(a.dtype) if center is not None else np.zeros_like(a3)
    for (dz, dy, dx), w in zip(offs, wts):
        iz, iy, ix = pad_index(zz + dz, Z, mode), pad_index(yy + dy, Y, mode), pad_index(xx + dx, X, mode)
        ok = (iz >= 0) & (iy >= 0) & (ix >= 0)
        v = np.where(ok, a3[np.maximum(iz, 0), np.maximum(iy, 0), np.maximum(ix, 0)], a.dtype.type(fill))
        acc = acc + v * a.dtype.type(w)
    return acc.reshape(a.shape)


@pytest.mark.parametrize("mode", sorted(_PAD_MODES))
@pytest.mark.parametrize("shape,kshape", [((40,), (4,)), ((23, 31), (3, 5)), ((17, 19), (4, 2)), ((5, 9, 11), (3, 3, 2))])
def test_reference_tap_list_reproduces_the_reference_loop(shape, kshape, mode):
    rng = np.random.default_rng(len(shape) * 100 + kshape[0])
    a = rng.standard_normal(shape)
    k = rng.random(kshape)
    for s in (0, 1, 2):
        kc, offs, wts = _reference_taps(k, s)
        got = apply_taps(a, kc, offs, wts, mode)
        np.testing.assert_allclose(got, O.atrous_convolution_nd(a, k, None, s, mode), rtol=0, atol=1e-12)


@pytest.mark.parametrize("taps", [[0.5, 0.5], [0.1, 0.4, 0.3, 0.2], list(np.hanning(19)[1:-1] / np.hanning(19)[1:-1].sum())])
def test_filter_tap_list_reproduces_convolution_for_even_and_long_taps(taps):
    rng = np.random.default_rng(len(taps))
    t = np.asarray(taps)
    for a in (rng.standard_normal(90), rng.standard_normal((29, 37)), rng.standard_normal((6, 9, 13))):
        kern = t
        for _ in range(a.ndim - 1):
            kern = np.multiply.outer(kern, t)
        for s in (0, 1, 2):
            offs, wts = _filter_taps(kern, s, convolve=a.ndim == 1)
            got = apply_taps(a, None, offs, wts, "reflect" if a.ndim == 1 else "symmetric")
            np.testing.assert_allclose(got, O.convolution_taps_nd(a, t, s), rtol=0, atol=1e-12)


class _Even4(__import__("wavelets_amd").wavelets.AbstractScalingFunction):        # (module level: picklable)
    coefficients_1d = np.array([0.1, 0.4, 0.3, 0.2])
    sigma_e_1d = sigma_e_2d = sigma_e_3d = np.array([0.9, 0.2, 0.09, 0.04])

    def __init__(self, *args, **kwargs):
        super().__init__("even4", *args, **kwargs)


def test_generic_tap_coefficients_reacquire_a_storage_plan_when_they_lost_theirs(monkeypatch):
    """Coefficients of a scaling function with an even number of taps (or more than 15) live on a
    storage-only plan of the generic operator.  A copy, an unpickled object or Coefficients(ndarray,
    sf) has no plan: the first device operation must acquire that storage plan again instead of
    asking the tuned engine for a family it does not have (_taps_f64 / _family_of raise
    NotImplementedError for such taps).  No GPU here: the plan pool is replaced by a recorder."""
    import copy
    import pickle
    from wavelets_amd import wavelets as WV

    Even4 = _Even4
    calls = []

    class FakePlan:
        nranks = 1

        def __init__(self, kind, *a):
            calls.append((kind,) + a)
            self.uploaded = []

        def upload(self, plane, host):
            self.uploaded.append((plane, host.shape, host.dtype))

    monkeypatch.setattr(WV, "default_context", lambda: "ctx")
    monkeypatch.setattr(WV, "acquire_plan", lambda ctx, H, W, fam, lvl: FakePlan("f32", H, W, fam, lvl))
    monkeypatch.setattr(WV, "acquire_plan64", lambda ctx, H, W, taps, lvl: FakePlan("f64", H, W, taps, lvl))
    monkeypatch.setattr(WV, "release_plan", lambda plan: None)
    for dtype, kind in ((np.float32, "f32"), (np.float64, "f64")):
        for shape, hw in (((3, 10, 12), (10, 12)), ((3, 40), (1, 40)), ((3, 4, 5, 6), (20, 6))):
            sf = Even4(len(shape) - 1)
            c = WV.Coefficients(np.zeros(shape, dtype), sf)
            for obj in (c, copy.deepcopy(c), copy.copy(c), pickle.loads(pickle.dumps(c))):
                del calls[:]
                plan = obj._device()                       # what denoise / significance / sum call first
                assert calls == [(kind,) + hw + ((WV._lib.B3SPLINE if kind == "f32" else (1.0,)), 2)], calls
                assert [u[0] for u in plan.uploaded] == [0, 1, 2] and plan.uploaded[0][1] == hw
                obj._plan = None


@pytest.mark.parametrize("taps", [[0.5, 0.5], [0.1, 0.4, 0.3, 0.2], list(np.hanning(19)[1:-1] / np.hanning(19)[1:-1].sum())])
def test_axis_by_axis_generic_filter_reproduces_convolution_and_the_recursive_base_operator(taps):
    """_generic_filter (round 4) applies a scaling function's outer-product kernel axis by axis through
    the generic operator - K taps per axis instead of K**ndim.  A numpy model of wt_taps_kernel stands in
    for the plan: (a) standard algorithm: the result is the oracle's convolution() for even / long taps
    at every scale; (b) recursive algorithm (offsets (j - K // 2) * d under the polyphase border rule):
    every polyphase sub-array equals the base operator applied to it on its own."""
    from wavelets_amd import wavelets as WV

    class SF:
        coefficients_1d = np.asarray(taps)

    class ModelPlan:
        def __init__(self, a):
            self.planes = {WV.PLANE_INPUT: a}
            self.nd = a.ndim

        def taps_conv(self, src, var, dst, offs, wts, kc, depth=0, pad_mode=0, fill_value=0.0, dilation=1):
            a = self.planes[src]
            if pad_mode in (WV._lib.PAD_POLY_SYMMETRIC, WV._lib.PAD_POLY_MIRROR):
                out = np.zeros_like(a)
                mode = "symmetric" if pad_mode == WV._lib.PAD_POLY_SYMMETRIC else "reflect"
                for starts in np.ndindex(*(dilation,) * a.ndim):          # every residue class on its own
                    sl = tuple(slice(st, None, dilation) for st in starts)
                    sub = a[sl]
                    if sub.size:
                        out[sl] = apply_taps(sub, None, np.asarray(offs) // dilation, wts, mode)
                self.planes[dst] = out
            else:
                mode = {v: k for k, v in _PAD_MODES.items()}[pad_mode]
                self.planes[dst] = apply_taps(a, None, offs, wts, mode)

        def axis_filter(self, src, dst, axis, offsets, weights, depth=0, pad_mode=0, fill_value=0.0, dilation=1):
            """wt_axis_filter: axis 2 = x, 1 = y, 0 = z of the (Z, Y, X) view - the last axes of the array"""
            offs = np.zeros((len(offsets), 3), dtype=np.int64)
            offs[:, axis] = offsets
            self.taps_conv(src, None, dst, offs, weights, None, depth, pad_mode, fill_value, dilation)

    rng = np.random.default_rng(len(taps))
    t = np.asarray(taps)
    for a in (rng.standard_normal(70), rng.standard_normal((23, 31)), rng.standard_normal((5, 9, 12))):
        for s in (0, 1, 2):
            plan = ModelPlan(a)
            WV._generic_smooth(plan, SF, a.ndim, a.shape, WV.PLANE_INPUT, WV.PLANE_OUT, s)
            np.testing.assert_allclose(plan.planes[WV.PLANE_OUT], O.convolution_taps_nd(a, t, s), rtol=0, atol=1e-12)
            d = 2 ** s
            plan = ModelPlan(a)
            WV._generic_filter(plan, SF, a.ndim, a.shape, WV.PLANE_INPUT, WV.PLANE_OUT, d, 0,
                               WV._lib.PAD_POLY_MIRROR if a.ndim == 1 else WV._lib.PAD_POLY_SYMMETRIC, dilation=d)
            want = np.zeros_like(a)
            for starts in np.ndindex(*(d,) * a.ndim):
                sl = tuple(slice(st, None, d) for st in starts)
                if a[sl].size:
                    want[sl] = O.convolution_taps_nd(np.ascontiguousarray(a[sl]), t, 0)
            np.testing.assert_allclose(plan.planes[WV.PLANE_OUT], want, rtol=0, atol=1e-12)


@pytest.mark.parametrize("shape,kshape", [((72, 100), (25, 23)), ((75, 100), (25, 23)), ((75, 100), (24, 26)),
                                          ((33, 20), (33, 19)), ((48, 40), (1, 7)), ((21, 6), (2, 2))])
def test_extended_frame_reproduces_the_circular_products_of_the_image(shape, kshape):
    """richardson_lucy(fft=True) on sides the engine's FFT does not take (utils._ExtendedFFT): the index
    logic - periodic extension windows, PSF laid around the frame's origin, window copied back -
    replayed with numpy's FFT on the frame must equal the reference's products on the image itself
    (utils.py:246-254, 284), odd heights included."""
    from wavelets_amd import utils as WU
    rng = np.random.default_rng(sum(shape) + sum(kshape))
    H, W = shape
    x = rng.standard_normal(shape)
    k = rng.uniform(0.1, 1.0, kshape)
    kh, kw = kshape
    # the reference's kernel spectrum (utils.py:246-250)
    pad = np.zeros(shape)
    pad[H // 2 - kh // 2:H // 2 - kh // 2 + kh, W // 2 - kw // 2:W // 2 - kw // 2 + kw] = k
    f = np.fft.rfft2(np.roll(pad, (H // 2, W // 2), axis=(0, 1)))
    want_conv = np.fft.irfft2(np.fft.rfft2(x) * f, s=shape)
    want_corr = np.fft.irfft2(np.fft.rfft2(x) * f.conj(), s=shape)
    e, hy, hx, Mh, Mw = WU._ext_geometry(H, W, kh, kw)
    def smooth(n):
        for r in (2, 3, 5):
            while n % r == 0:
                n //= r
        return n == 1
    # (round 5: the frame's sides are the next products of 2s, 3s and 5s - what wt_fft.h transforms)
    assert smooth(Mh) and smooth(Mw) and H + 2 * hy <= Mh < 2 * (H + 2 * hy) and W + 2 * hx <= Mw < 2 * (W + 2 * hx)
    assert not any(smooth(m) for m in range(H + 2 * hy, Mh)) and not any(smooth(m) for m in range(W + 2 * hx, Mw))
    F = np.fft.fft2(WU._ext_kernel_frame(k, H, Mh, Mw))
    frame = np.zeros((Mh, Mw))
    for sy, sx, dy, dx, nr, nc in WU._ext_windows(H, W, hy, hx):
        frame[dy:dy + nr, dx:dx + nc] = x[sy:sy + nr, sx:sx + nc]
    X = np.fft.fft2(frame)
    conv = np.fft.ifft2(X * F).real[hy:hy + H, hx:hx + W]
    corr = np.fft.ifft2(X * F.conj()).real[hy:hy + H, hx:hx + W]
    np.testing.assert_allclose(conv, want_conv, atol=1e-11)
    np.testing.assert_allclose(corr, want_corr, atol=1e-11)


def test_extension_windows_tile_the_extended_image_exactly_once():
    """utils._ext_windows: the device copies that extend an H x W image periodically by (hy, hx) must write
    every element of the (H + 2 hy) x (W + 2 hx) rectangle exactly once, each from the pixel
    ((y - hy) mod H, (x - hx) mod W) - for halos from 0 up to the full image size."""
    from wavelets_amd import utils as WU
    rng = np.random.default_rng(5)
    for _ in range(200):
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        hy, hx = int(rng.integers(0, H + 1)), int(rng.integers(0, W + 1))
        src = np.arange(H * W).reshape(H, W)
        dst = np.full((H + 2 * hy, W + 2 * hx), -1)
        hits = np.zeros_like(dst)
        for sy, sx, dy, dx, nr, nc in WU._ext_windows(H, W, hy, hx):
            assert nr > 0 and nc > 0 and sy >= 0 and sx >= 0 and sy + nr <= H and sx + nc <= W
            dst[dy:dy + nr, dx:dx + nc] = src[sy:sy + nr, sx:sx + nc]
            hits[dy:dy + nr, dx:dx + nc] += 1
        assert (hits == 1).all()
        yy, xx = np.mgrid[0:H + 2 * hy, 0:W + 2 * hx]
        assert np.array_equal(dst, src[(yy - hy) % H, (xx - hx) % W])


def test_fft_sizes_host_logic_matches_the_five_smooth_rule():
    """wt_fft_supported (host logic of the C library, no GPU): exactly the sides 2 .. 8192 without a prime factor
    above 5 - what utils._next_smooth rounds the extended frame up to."""
    from wavelets_amd import _lib as L
    from wavelets_amd import utils as WU

    def smooth(n):
        for r in (2, 3, 5):
            while n % r == 0:
                n //= r
        return n == 1
    for n in list(range(1, 700)) + [1000, 1024, 3000, 3072, 3125, 6075, 6561, 7776, 8000, 8191, 8192, 8193, 8748, 9000, 16384]:
        want = 2 <= n <= 8192 and smooth(n)
        assert L.fft_supported(n, 64) == want and L.fft_supported(60, n) == want, n
    for n in (1, 2, 7, 11, 97, 121, 1025, 3066 + 64, 4097, 8000):
        m = WU._next_smooth(n)
        assert m >= max(n, 2) and smooth(m) and not any(smooth(k) for k in range(max(n, 2), m)), (n, m)
