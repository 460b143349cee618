"""Host-side mirror of ``watroo.utils``: ``denoise`` and ``wow`` on the MI355X engine.

Same signatures and plumbing as /root/reference/watroo/utils.py (cited ``ref:LINE``); the
planes never leave HBM between the transform and the final reconstruction download.
``richardson_lucy`` (ref:222-290) keeps its whole iteration
on the device.
"""
import copy
import warnings

import numpy as np

from . import _lib
from ._lib import PLANE_NONE, PLANE_OUT, PLANE_SCRATCH, acquire_plan, acquire_plan64, default_context, release_plan
from .wavelets import (AtrousTransform, B3spline, Coefficients, _decompose_denoise_sum, _f32_source, _f64_source, _family_of, _needs_generic,
                       _result_dtype, _taps_f64, _to_f32_image,
                       generalized_anscombe,
                       PLANE_INPUT)

__all__ = ['denoise', 'wow', 'richardson_lucy']     # enhance / prepare_params importable by path, as in the reference

_POWER_PLANE = PLANE_SCRATCH(3)
_GAMMA_PLANE = PLANE_SCRATCH(4)
_SQ_PLANE, _POW_PLANE = PLANE_SCRATCH(6), PLANE_SCRATCH(7)     # 3-D wow: c^2 and its 3-D smoothing
# wow behind a bilateral transform: all but this many planes are summed early, beside the transform's last scales
# (8192^2, 12 planes, tools/ab_sumtail.sh: float32 6.32 ms with the sum in one piece, 6.46 / 6.27 / 6.34 with 3 / 4 / 5
#  planes left for the end; float64 16.39, 16.56 / 16.28 / 16.06)
_SUM_TAIL_ENV = __import__("os").environ.get("WATROO_HIP_SUM_TAIL")
_SUM_TAIL_PLANES = int(_SUM_TAIL_ENV) if _SUM_TAIL_ENV else 4
_SUM_TAIL_PLANES_F64 = int(_SUM_TAIL_ENV) if _SUM_TAIL_ENV else 5


def prepare_params(param, ndims):
    """Normalise a scalar / list / per-channel list parameter (ref:10-33)."""
    if ndims == 2:
        if param is None:
            return []
        return copy.copy(param) if type(param) is list else [param]
    if type(param) is not list:
        return [[], ] * ndims if param is None else [[param], ] * ndims
    if len(param) != ndims:
        raise ValueError("Invalid number of parameters")                  # ref:26
    out = [prepare_params(p, 2) for p in param]
    if None in out:
        out[out.index(None)] = []
    return out


def enhance(*args, weights=None, denoise=None, soft_threshold=True, out=None, **kwargs):
    """De-noising and / or enhancement by modification of the wavelet coefficients, per channel
    for (3, H, W) images (ref:36-80): transform over ``len(weights)`` scales, threshold with
    ``denoise`` sigmas, recombine with ``weights``.  ``args = (img[, noise])``; ``kwargs`` go to
    ``AtrousTransform``."""
    img = np.asarray(args[0])
    channels = [0, 1, 2] if img.ndim == 3 else [Ellipsis]                 # ref:47-50
    if out is None:
        out = _lib.host_empty(img.shape, dtype=_result_dtype(img))
    weights = prepare_params(weights, img.ndim)
    denoise = prepare_params(denoise, img.ndim)
    atrous = AtrousTransform(**kwargs)
    plans = []
    for c in channels:                                                    # the reference's list plumbing, in its order
        dns = denoise if c is Ellipsis else denoise[c]
        wgt = weights if c is Ellipsis else weights[c]
        if len(wgt) < len(dns):                                           # ref:65-68
            wgt.extend([1] * (len(dns) - len(wgt)))
        elif len(dns) < len(wgt):
            dns.extend([0] * (len(wgt) - len(dns)))
        plans.append((c, dns, wgt))

    def channel(item):
        c, dns, wgt = item
        coeffs = atrous(img[c], len(wgt))                                 # ref:70
        if len(args) == 2:
            coeffs.noise = args[1] if c is Ellipsis else args[1][c]       # ref:71-72
        else:
            coeffs.noise = coeffs.get_noise()
        plan = coeffs._denoise_sum(dns, weights=wgt, soft_threshold=soft_threshold,
                                   write_back=False)                       # ref:76-78
        tgt = out[c]
        want = np.float64 if isinstance(plan, _lib.Plan64) else np.float32
        if (isinstance(tgt, np.ndarray) and tgt.ndim == 2 and tgt.dtype == want and tgt.shape == tuple(plan.shape)
                and tgt.strides[1] == tgt.itemsize and tgt.flags.writeable):
            plan.download(PLANE_OUT, out=tgt)                             # straight into the caller's rows
        else:
            out[c] = plan.download(PLANE_OUT)
        return None

    # the channels of a colour image are independent frames: one lane each, so that the upload of one
    # channel, the passes of another and the download of a third overlap (sequence.map_frames; bit-identical)
    # (from a megapixel per channel: below that the PCIe legs are microseconds, and the lanes' contexts - created on
    #  their first use in a process, ~0.1 s each - would cost a one-shot call more than they can ever save it)
    from .sequence import map_frames
    big = len(plans) > 1 and img[0].size >= (1 << 20)
    map_frames(channel, plans, lanes=len(plans) if big else 1)
    return out


def _denoise_pipelined(plan, img, level, sf, weights, noise, bilateral, soft_threshold, anscombe, out=None):
    """denoise() with the noise level GIVEN as a scalar: every threshold is known before the first pixel
    arrives, so the whole call is one pipelined host-to-host pass (wt_denoise_sum_host: upload, passes,
    thresholds, passes, download over blocks of rows - about one PCIe leg instead of two).  Returns the
    result, or None when the case is not the pipeline's (the caller runs the serial sequence)."""
    if noise is None or np.ndim(noise) != 0 or bilateral is not None or anscombe or plan.custom or level < 2:
        return None
    if img.size < (1 << 22) or not plan.fused_ok(level):
        return None
    entries = list(zip(range(level + 1), weights, (1,) * len(weights)))
    n_den = max([scl + 1 for scl, sig, _ in entries if sig != 0], default=0)
    sched = _lib.schedule(plan.family, level, True)
    k, covered = 0, 0
    while k < len(sched) and (covered < n_den or k == 0):
        covered += sched[k][1]
        k += 1
    if k == 0 or k >= len(sched) or n_den == 0:
        return None
    sigma_e = sf.sigma_e()
    taus = []
    for scl in range(covered):
        sig = weights[scl] if scl < len(weights) else 0
        tau = float(sig * noise * sigma_e[scl]) if sig != 0 and noise != 0 else 0.0     # ref wavelets.py:133-141
        if tau < 0:                      # erf(|w / tau|) = erf(|w| / |tau|); |w| > tau always true (hard)
            tau = -tau if soft_threshold else 0.0
        taus.append(tau)
    try:
        if not (isinstance(out, np.ndarray) and out.ndim == 2 and out.dtype == np.float32 and out.shape == tuple(plan.shape)
                and out.strides[1] == 4 and out.flags.writeable):
            out = None
        return plan.denoise_sum_host(img, level, k, taus, [1.0] * covered, soft_threshold, out=out)
    except _lib.WatrooHipError as e:
        if "wt_denoise_sum_host: the threshold step" in str(e):     # no pipeline for this plan / size / option
            return None
        raise


def _download_to(plan, target):
    """PLANE_OUT of `plan` as a host array: straight into `target` when it is a writable 2-D array of the plan's
    shape and element type with contiguous rows (sequence.denoise_many(out=...): no intermediate copy), else a fresh
    page-locked block."""
    want = np.float64 if isinstance(plan, _lib.Plan64) else np.float32
    if (isinstance(target, np.ndarray) and target.ndim == 2 and target.dtype == want and target.shape == tuple(plan.shape)
            and target.strides[1] == target.itemsize and target.flags.writeable):
        return plan.download(PLANE_OUT, out=target)
    return plan.download(PLANE_OUT)


def denoise(data, weights, scaling_function=B3spline, noise=None, bilateral=None,
            soft_threshold=True, anscombe=False, *, _out=None):
    """Denoise ``data``: transform over ``len(weights)`` scales, threshold each scale at
    ``weights[s]`` sigma, sum the planes (ref:83-102).  Optional Anscombe pre/post transform.
    Everything between the upload of ``data`` and the download of the result runs on the GPU.
    """
    f64 = _result_dtype(data) == np.float64                           # float64 engine (ref:319-320)
    if f64 and np.ndim(data) == 2 and bilateral is None and not _needs_generic(scaling_function):
        # float64 images: the same interleaving as float32 below on the float64 engine - fused passes,
        # the first one histogramming |w_0|, thresholds + start of the sum in one kernel
        # (wt64_denoise_sum), the later passes carrying the sum
        img = _f64_source(data)           # (integer images are widened on the device)
        level = len(weights)
        transform = AtrousTransform(scaling_function)
        sf = scaling_function(2)
        plan = acquire_plan64(default_context(), img.shape[0], img.shape[1], _taps_f64(sf, 2), level)
        plan.upload(PLANE_INPUT, img)
        if anscombe:
            plan.anscombe(PLANE_INPUT, PLANE_INPUT)                       # ref:93-94
        coefficients = Coefficients(plan, sf, None)
        coefficients.noise = noise                                        # ref:96
        _decompose_denoise_sum(transform, plan, level, coefficients, weights,
                               soft_threshold=soft_threshold, write_back=False)
        if anscombe:
            plan.anscombe(PLANE_OUT, PLANE_OUT, inverse=True)             # ref:99-100
        return _download_to(plan, _out)
    if np.ndim(data) in (1, 3) or (f64 and np.ndim(data) == 2) or _needs_generic(scaling_function):
        # signals, cubes and float64 images: the generic call sequence
        arr = np.asarray(data, np.float64 if f64 else np.float32)
        if anscombe:
            arr = generalized_anscombe(arr)
        coefficients = AtrousTransform(scaling_function, bilateral=bilateral)(arr, len(weights))
        coefficients.noise = noise
        plan = coefficients._denoise_sum(weights, soft_threshold=soft_threshold, write_back=False)
        if anscombe:
            plan.anscombe(PLANE_OUT, PLANE_OUT, inverse=True)
        return coefficients._from_plane(plan.download(PLANE_OUT)).astype(_result_dtype(data), copy=False)
    img = _f32_source(data, "data")       # (integer / byte-swapped images are widened on the device)
    level = len(weights)
    transform = AtrousTransform(scaling_function, bilateral=bilateral)
    sf = scaling_function(2)
    plan = acquire_plan(default_context(), img.shape[0], img.shape[1], _family_of(sf), level)
    piped = None
    if img.dtype == np.float32:           # the pipelined host call takes float32 rows
        piped = _denoise_pipelined(plan, img, level, sf, weights, noise, bilateral, soft_threshold, anscombe, out=_out)
    if piped is not None:
        release_plan(plan)
        return piped.astype(_result_dtype(data), copy=False)
    plan.upload(PLANE_INPUT, img)
    if anscombe:
        plan.anscombe(PLANE_INPUT, PLANE_INPUT)                           # ref:93-94
    coefficients = Coefficients(plan, sf, bilateral)
    coefficients.noise = noise                                            # ref:96
    # ref:95, 97-98: transform, threshold and sum interleaved (trailing zero sigmas - scales that
    # are transformed but not thresholded - ride on the accumulate passes); the thresholded planes
    # themselves are not returned by denoise(), so they are not written back
    _decompose_denoise_sum(transform, plan, level, coefficients, weights,
                           soft_threshold=soft_threshold, write_back=False)
    if anscombe:
        plan.anscombe(PLANE_OUT, PLANE_OUT, inverse=True)                 # ref:99-100
    return _download_to(plan, _out).astype(_result_dtype(data), copy=False)


def _pad_list(values, n, fill):
    out = copy.copy(values)
    if len(out) <= n:
        out.extend([fill, ] * (n - len(out) + 1))
    return out


def wow(data,
        scaling_function=B3spline,
        n_scales=None,
        weights=[],
        whitening=True,
        denoise_coefficients=[],
        noise=None,
        bilateral=None,
        bilateral_scaling=False,
        soft_threshold=True,
        preserve_variance=False,
        gamma=3.2,
        gamma_min=None,
        gamma_max=None,
        h=0):
    """Wavelets Optimized Whitening (ref:105-219; Auchere et al. 2023).

    ``data`` is a 2-D ndarray or a ``Coefficients`` object (which is then mutated and
    returned, ref:128-131,152-153).  Returns ``(image, coefficients)``; the coefficients are
    the whitened ones, exactly as in the reference.
    """
    if type(data) is np.ndarray:                                          # ref:121-127
        if data.ndim > 3:
            raise ValueError("Unsupported number of dimensions")
        max_scales = int(np.round(np.log2(min(data.shape))
                                  - np.log2(len(scaling_function.coefficients_1d))))
        if n_scales is None:
            n_scales = max_scales if h < 1 else len(denoise_coefficients)
        elif n_scales > max_scales:
            n_scales = max_scales
        n_dims = data.ndim
    elif type(data) is Coefficients:                                      # ref:128-131
        n_scales = len(data) - 1
        n_dims = data._ndim
        scaling_function = data.scaling_function.__class__
    else:
        raise ValueError('Unknown input type')                            # ref:133

    max_scales = len(scaling_function(n_dims).sigma_e(bilateral=bilateral))
    if len(denoise_coefficients) >= max_scales:                           # ref:136-138
        warnings.warn('Required number of scales lager then the maximum for scaling '
                      f'function. Using {max_scales}.')
        n_scales = max_scales

    if bilateral is None:                                                 # ref:140-146
        sigma_bilateral = None
    else:
        sigma_bilateral = copy.copy(bilateral) if type(bilateral) is list \
            else [bilateral, ] * (n_scales + 1)
        if len(sigma_bilateral) <= n_scales:
            sigma_bilateral.extend([1, ] * (n_scales - len(sigma_bilateral) + 1))

    if type(data) is np.ndarray:                                          # ref:148-151
        transform = AtrousTransform(scaling_function, bilateral=sigma_bilateral,
                                    bilateral_scaling=bilateral_scaling)
        coefficients = transform(data, n_scales)      # float64 engine for float64 data, no bilateral
        coefficients.noise = noise
    else:
        coefficients = data

    plan = _wow_device(coefficients, n_scales, weights, whitening, denoise_coefficients,
                       soft_threshold, preserve_variance, gamma, gamma_min, gamma_max, h)
    nplanes = len(coefficients)
    recon = coefficients._from_plane(plan.download(PLANE_OUT)).astype(coefficients._dtype, copy=False)
    coefficients._refresh_host(range(nplanes))
    return recon, coefficients


def _wow_device(coefficients, n_scales, weights, whitening, denoise_coefficients, soft_threshold,
                preserve_variance, gamma, gamma_min, gamma_max, h):
    """The device-resident part of wow (ref:157-217): per-scale loop, plane sum, gamma blend.
    Leaves the whitened planes on the plan and the image in PLANE_OUT; returns the plan.
    (bench.py --config cfg5 times exactly this behind the transform, without the PCIe legs.)"""
    plan = coefficients._device()
    coefficients._sum_valid = False
    npix = float(plan.H) * float(plan.W)

    use_gamma = h > 0
    if use_gamma:
        plan.fill(_GAMMA_PLANE, 0.0)                                      # ref:157-158

    recomposition_weights = _pad_list(weights, n_scales, 1)               # ref:160-163
    sdc = copy.copy(denoise_coefficients)                                 # ref:165-170
    if len(sdc) < n_scales:
        sdc.extend([0, ] * (n_scales - len(sdc)))
    if len(sdc) == n_scales:
        sdc.extend([1, ])

    nplanes = len(coefficients)
    gplane = _GAMMA_PLANE if use_gamma else PLANE_NONE
    if coefficients._ndim == 1:
        plan.set_border(2)        # the 1-D branch filters with scipy's 'mirror' border (ref:65-69)
    try:
        early = _wow_scales(plan, coefficients, n_scales, nplanes, recomposition_weights, sdc, npix,
                            preserve_variance, whitening, h, soft_threshold, gplane)
    finally:
        plan.set_border(0)

    if early:                                                             # ref:205, the rest of it
        plan.plane_sum_resume(early, nplanes - early, PLANE_OUT)
    else:
        plan.plane_sum(0, nplanes, PLANE_OUT)                             # ref:205

    if use_gamma:                                                         # ref:207-217
        if gamma_min is None or gamma_max is None:
            _, _, lo, hi = plan.reduce(_GAMMA_PLANE)
            if gamma_min is None:
                gamma_min = lo
            if gamma_max is None:
                gamma_max = hi
        plan.gamma_blend(PLANE_OUT, _GAMMA_PLANE, gamma_min, gamma_max, 1 / gamma, h)

    return plan


def _wow_scales(plan, coefficients, n_scales, nplanes, recomposition_weights, sdc, npix,
                preserve_variance, whitening, h, soft_threshold, gplane):
    """The per-scale loop of wow (ref:174-203).  Returns the number of leading planes whose sum (ref:205) is
    already queued into PLANE_OUT (0: none) - behind a bilateral transform the planes of the first scales are
    final while the transform's last scales still run, and their share of the sum goes beside those
    (Plan.plane_sum_early; identical bits, the additions keep their order)."""
    ft = np.float64 if isinstance(plan, _lib.Plan64) else np.float32      # the data's compute type
    tail = _SUM_TAIL_PLANES_F64 if ft is np.float64 else _SUM_TAIL_PLANES
    early, early_at = 0, (nplanes - tail if nplanes >= 2 * tail else 0)
    for s, (_, w, d) in enumerate(zip(range(nplanes), recomposition_weights, sdc)):  # ref:174
        if s and s == early_at and hasattr(plan, "plane_sum_early") and plan.plane_sum_early(s, PLANE_OUT):
            early = s
        need_moments = preserve_variance or (s == n_scales and whitening and h < 1)
        if need_moments:
            tot, tot2, _, _ = plan.reduce(s)
            mean = tot / npix
            std = ft(np.sqrt(max(tot2 / npix - mean * mean, 0.0)))
            rms = ft(np.sqrt(tot2 / npix))
        if preserve_variance:                                             # ref:178-184
            power_norm = std if s == n_scales else rms
        else:
            power_norm = 1
        if s == n_scales:                                                 # ref:185-191
            if whitening and h < 1:
                local_power = std
                if local_power <= 0:
                    local_power = 1e-15
            else:
                local_power = 1
            factor = ft(w * power_norm / local_power)                     # ref:203
            plan.wow_update(s, PLANE_NONE, 0.0, soft_threshold, PLANE_NONE, factor, gplane)
        else:
            t = coefficients._tau(d, s, soft_threshold)                   # ref:199
            tau, noise_plane = (0.0, PLANE_NONE) if t is None else t
            factor = ft(w * power_norm)
            if whitening and h < 1 and coefficients._ndim == 3:           # ref:193-196 on a cube
                # local power = 3-D conv_s(c^2): per-slice 2-D filter + axis-0 filter
                plan.binary("mul", s, s, _SQ_PLANE)
                plan.smooth3d(_SQ_PLANE, _POW_PLANE, s, coefficients._shape[0])
                plan.wow_update(s, _POW_PLANE, tau, soft_threshold, noise_plane, factor, gplane)
            elif (whitening and h < 1 and isinstance(plan, _lib.Plan64) and coefficients._ndim == 2
                  and not _needs_generic(coefficients.scaling_function)):
                # float64 images: row pass of the squares, column pass with the update as its epilogue
                plan.wow_scale(s, s, tau, soft_threshold, noise_plane, factor, gplane)
            elif whitening and h < 1 and plan.custom:                     # user-defined taps
                plan.smooth(s, _POW_PLANE, s, True)                       # conv_s(c^2), ref:194
                plan.wow_update(s, _POW_PLANE, tau, soft_threshold, noise_plane, factor, gplane)
            elif whitening and h < 1:                                     # ref:193-196 + 199-203
                # local power conv_s(c^2), significance, gamma sum and whitening in one kernel
                plan.wow_scale(s, s, tau, soft_threshold, noise_plane, factor, gplane)
            else:
                plan.wow_update(s, PLANE_NONE, tau, soft_threshold, noise_plane, factor, gplane)
    return early


def _periodic_operand(kernel, ay, ax):
    """(kernel, filter2d keywords) of a periodic correlation anchored at row ay, which may fall
    just outside the kernel (odd image heights, one-row PSFs): zero rows are added until the
    anchor lies inside - wt_filter2d_ex wants 0 <= ay < kh."""
    kh = kernel.shape[0]
    if ay < 0:
        kernel = np.concatenate([np.zeros((-ay, kernel.shape[1]), kernel.dtype), kernel])
        ay = 0
    elif ay >= kh:
        kernel = np.concatenate([kernel, np.zeros((ay - kh + 1, kernel.shape[1]), kernel.dtype)])
    return np.ascontiguousarray(kernel), dict(anchor=(ay, ax), periodic=True)


# PSFs of at least this many taps take the FFT form of the circular products (images whose sides are
# products of 2s, 3s and 5s; other sizes: times twice the area ratio of the frame they are extended into)
_FFT_MIN_TAPS = int(__import__("os").environ.get("WATROO_HIP_FFT_MIN_TAPS", "512"))
# tests: take the extended frame even where the image's own sides qualify for the FFT
_FFT_FORCE_EXTENDED = False


def _next_smooth(n):
    """the smallest m >= max(n, 2) without a prime factor above 5 (the lengths wt_fft.h transforms)"""
    m = max(2, int(n))
    while True:
        r = m
        for f in (2, 3, 5):
            while r % f == 0:
                r //= f
        if r == 1:
            return m
        m += 1


def _ext_geometry(H, W, kh, kw):
    """(e, hy, hx, Mh, Mw) of the extended frame: halo rows / columns and the frame size (5-smooth sides)"""
    e = H % 2
    hy, hx = kh // 2 + e, kw // 2
    return e, hy, hx, _next_smooth(H + 2 * hy), _next_smooth(W + 2 * hx)


def _ext_kernel_frame(psf, H, Mh, Mw):
    """The PSF laid around the frame's origin as ref utils.py:246-250 lays it around the image's"""
    kh, kw = psf.shape
    frame = np.zeros((Mh, Mw), dtype=psf.dtype)
    rows = (np.arange(kh) - kh // 2 - H % 2) % Mh
    cols = (np.arange(kw) - kw // 2) % Mw
    frame[np.ix_(rows, cols)] = psf
    return frame


def _ext_windows(H, W, hy, hx):
    """(sy, sx, dy, dx, rows, cols) copies that extend an H x W image periodically by (hy, hx)"""
    return [(sy, sx, dy, dx, nr, nc)
            for sy, dy, nr in ((H - hy, 0, hy), (0, hy, H), (0, hy + H, hy))
            for sx, dx, nc in ((W - hx, 0, hx), (0, hx, W), (0, hx + W, hx)) if nr and nc]


class _ExtendedFFT:
    """Circular products of an H x W image with a side the engine's FFT does not take (a prime factor
    above 5; wt_fft.h transforms lengths 2^a 3^b 5^c) through that FFT.  A product with a kernel of
    support [-hy, hy] x [-hx, hx] only looks
    hy rows / hx columns beyond a pixel, so the image is extended PERIODICALLY by that much on every
    side (nine device window copies), placed in a zeroed frame - the next 5-smooth sizes that hold H + 2 hy rows and
    W + 2 hx columns, the frame's own circular product is taken, and the H x W window is copied back:
    inside it no term has wrapped around the frame, so it equals the product of period (H, W).  The
    kernel spectrum is that of the PSF laid around the frame's origin exactly as the reference lays it
    around the image's (ref utils.py:246-250; for an odd height the reference's two rolls by H // 2
    leave the centre one row above the origin - `e`)."""

    def __init__(self, plan, f64, psf):
        H, W = plan.H, plan.W
        self.e, self.hy, self.hx, self.Mh, self.Mw = _ext_geometry(H, W, *psf.shape)
        self.plan, self.big = plan, None
        self.ok = self.hy <= H and self.hx <= W and _lib.fft_supported(self.Mh, self.Mw)
        self.f64, self.psf = f64, psf

    def worth_it(self, min_taps):
        kh, kw = self.psf.shape
        return self.ok and kh * kw * self.plan.H * self.plan.W >= 2 * min_taps * self.Mh * self.Mw

    def prepare(self, sf):
        ft = np.float64 if self.f64 else np.float32
        ctx = default_context()
        if self.f64:
            self.big = _lib.acquire_plan64(ctx, self.Mh, self.Mw, _taps_f64(sf, 2), 1)
        else:
            self.big = acquire_plan(ctx, self.Mh, self.Mw, _family_of(sf), 1)
        frame = _ext_kernel_frame(self.psf.astype(ft, copy=False), self.plan.H, self.Mh, self.Mw)
        self.A, self.B = PLANE_SCRATCH(0), PLANE_SCRATCH(1)
        self.big.upload(self.A, frame)
        self.big.fft_spectrum(self.A)
        self.big.fill(self.A, 0.0)             # the frame outside the extended image stays zero from here on

    def apply(self, src, dst, conj):
        H, W = self.plan.H, self.plan.W
        for sy, sx, dy, dx, nr, nc in _ext_windows(H, W, self.hy, self.hx):
            self.big.copy_window_from(self.plan, src, self.A, sy, sx, dy, dx, nr, nc)
        self.big.fft_apply(self.A, self.B, conj)
        self.plan.copy_window_from(self.big, self.B, dst, self.hy, self.hx, 0, 0, H, W)

    def close(self):
        if self.big is not None:
            _lib.release_plan(self.big)
            self.big = None


def richardson_lucy(data, psf,
                    iterations=10, denoise_coefficients=(5, 2, 1),
                    threshold_type='soft', uniform_init=False, persistent_mrs=True, fft=False):
    """Wavelet-regularised Richardson-Lucy deconvolution (ref:222-290), device-resident across
    iterations: per iteration one PSF correlation (ref:257), the residual's a-trous transform
    (ref:261), the multiresolution-support update per scale (ref:263-276), the plane sum
    (ref:278) and the second PSF correlation (ref:286) - nothing returns to the host until
    the final estimate.  ``fft=True`` selects the reference's circular (periodic-border)
    products (ref:245-254, 284): through the engine's own mixed-radix FFT for PSFs of 512 taps or more
    (wt_fft_apply; sides with a prime factor above 5: on a periodically extended frame), else as direct
    periodic correlations of the PSF (equal to the
    rFFT products up to rounding) - either way the loop stays on the device."""
    # float64 / promoted data: the float64 engine (ref wavelets.py:319-320) - except with
    # uniform_init, where the reference itself keeps the estimate in float32 (ref:233)
    f64 = _result_dtype(data) == np.float64 and not uniform_init and np.ndim(data) == 2
    ft = np.float64 if f64 else np.float32
    img = _f64_source(data) if f64 else _f32_source(data, "data")
    psf = np.ascontiguousarray(psf, dtype=ft)
    if psf.ndim != 2:
        raise ValueError("psf must be 2-D")
    level = len(denoise_coefficients)
    soft = threshold_type == 'soft'
    sf = B3spline(2)                                                     # ref:229 default transform
    if f64:
        plan = _lib.acquire_plan64(default_context(), img.shape[0], img.shape[1], _taps_f64(sf, 2), level)
    else:
        plan = acquire_plan(default_context(), img.shape[0], img.shape[1], _family_of(sf), level)
    DATA, PSI, PHI, RES, CONV = (PLANE_SCRATCH(i) for i in (6, 7, 8, 9, 10))
    if level > _lib.NUM_SCRATCH - 16:            # one support plane per scale (scratch 16 ..)
        raise ValueError(f"richardson_lucy: at most {_lib.NUM_SCRATCH - 16} denoise coefficients")
    MRS = [PLANE_SCRATCH(16 + s) for s in range(level)]
    plan.upload(DATA, img)
    plan.decompose(DATA, level)                                          # ref:230
    coefficients = Coefficients(plan, sf)
    if uniform_init:                                                     # ref:232-234
        tot = plan.reduce(DATA)[0]
        plan.fill(PSI, np.float32(np.float32(tot) / img.size))
    else:                                                                # ref:236-237
        coefficients._denoise_sum(list(denoise_coefficients), soft_threshold=soft,
                                  write_back=True)
        plan.copy(PLANE_OUT, PSI)
    for m in MRS:                                                        # ref:240-243
        plan.fill(m, 1.0 if soft else 0.0)
    psf_flipped = np.ascontiguousarray(psf[::-1, ::-1])
    kh, kw = psf.shape
    # fft=True: psi (*) psf circular with the PSF centre psf.shape // 2 at the origin (ref:246-250)
    # = periodic correlation with the flipped PSF anchored at k - 1 - k // 2; the second product
    # with conj(fft_psf) (ref:284) = periodic correlation with the PSF anchored at k // 2.  For an
    # ODD image height the reference's two rolls by H // 2 leave the PSF centre one row above the
    # origin, which moves the row anchors by one (g17_rl_fft_odd.npz); odd widths are not valid
    # in the reference (irfft2 returns W - 1 columns).
    if fft and img.shape[1] % 2:
        raise ValueError("richardson_lucy(fft=True) needs an even image width (numpy.fft.irfft2 "
                         "returns W - 1 columns in the reference)")
    e = img.shape[0] % 2
    # Large PSFs on images with 5-smooth sides: the products run through the engine's own FFT (wt_fft_apply:
    # row FFTs in LDS, transposes, the spectrum product fused into the first inverse pass) instead of
    # the direct periodic form, which costs kh * kw taps per pixel.  The periodic kernel image is built
    # as the reference builds it (ref:246-250) and transformed once.
    own_sides = _lib.fft_supported(img.shape[0], img.shape[1]) and not _FFT_FORCE_EXTENDED
    use_fft = bool(fft) and kh * kw >= _FFT_MIN_TAPS and kh <= img.shape[0] and kw <= img.shape[1] and own_sides
    ext = None
    if fft and not use_fft and kh <= img.shape[0] and kw <= img.shape[1] and not own_sides:
        # a side with a prime factor above 5: the same FFT on a periodically extended frame
        ext = _ExtendedFFT(plan, f64, psf)
        if ext.worth_it(_FFT_MIN_TAPS):
            ext.prepare(sf)
        else:
            ext = None
    if ext is not None:
        fwd_k = bwd_k = None
        fwd = bwd = {}
    elif use_fft:
        H, W = img.shape
        padded_psf = np.zeros((H, W), dtype=ft)
        padded_psf[H // 2 - kh // 2:H // 2 - kh // 2 + kh, W // 2 - kw // 2:W // 2 - kw // 2 + kw] = psf     # ref:247-249
        plan.upload(CONV, np.roll(padded_psf, (H // 2, W // 2), axis=(0, 1)))                                    # ref:250
        plan.fft_spectrum(CONV)
        fwd_k = bwd_k = None
        fwd = bwd = {}
    else:
        fwd_k, fwd = _periodic_operand(psf_flipped, kh - 1 - kh // 2 - e, kw - 1 - kw // 2) if fft \
            else (psf_flipped, {})
        bwd_k, bwd = _periodic_operand(psf, kh // 2 + e, kw // 2) if fft else (psf, {})
    data_noise = coefficients.noise       # None with uniform_init: every iteration then estimates
    try:
        for iteration in range(iterations):                                  # ref:252
            if ext is not None:
                ext.apply(PSI, PHI, False)                                   # ref:254
            elif use_fft:
                plan.fft_apply(PSI, PHI, False)                              # ref:254
            else:
                plan.filter2d(PSI, PHI, fwd_k, **fwd)                        # ref:255-257
            plan.binary("sub", DATA, PHI, RES)                               # ref:259
            plan.decompose(RES, level)                                       # ref:261
            # ref:262: a fresh Coefficients per iteration inherits the data's noise; when that is None
            # its lazy MAD estimate (ref:131-132) comes from the RESIDUAL's plane 0 - the planes the
            # plan holds right now.  One owner of the plan throughout: `coefficients`.
            coefficients.noise = data_noise
            for s, c in enumerate(denoise_coefficients):                     # ref:263-276
                t = coefficients._tau(c, s, soft)
                tau, noise_plane = (0.0, PLANE_NONE) if t is None else t
                plan.mrs_update(s, MRS[s], tau, soft, noise_plane, persistent_mrs,
                                1.0 / (iteration + 1))
            plan.plane_sum(0, level + 1, RES)                                # ref:278
            plan.binary("add_div", RES, PHI, RES)                            # ref:280-281
            if ext is not None:
                ext.apply(RES, CONV, True)                                   # ref:284
            elif use_fft:
                plan.fft_apply(RES, CONV, True)                              # ref:284
            else:
                plan.filter2d(RES, CONV, bwd_k, **bwd)                       # ref:284-286
            plan.binary("mul", PSI, CONV, PSI)                               # ref:288
    finally:
        if ext is not None:
            ext.close()
    # psi is float32 by construction with uniform_init (ref:233), else np.sum of the data's planes
    return plan.download(PSI).astype(np.float32 if uniform_init else _result_dtype(data), copy=False)
