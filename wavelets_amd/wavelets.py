"""Host-side mirror of ``watroo.wavelets`` for the MI355X engine.

Same public names, argument meaning and error behaviour as the reference module
(/root/reference/watroo/wavelets.py, cited as ``ref:LINE``), but the arithmetic runs in
``libwatroo_hip.so`` on the GPU and coefficient planes stay resident in HBM.

dtype policy: the reference keeps float64 inputs in float64 and promotes
int / big-endian inputs to float64 (ref:297,319-320; the README examples are float64).  Here the
tuned engine is float32 and serves float32 inputs; float64 / promoted inputs run on the float64
engine (``_lib.Plan64``: double planes, double arithmetic, generic kernels): the transform
(standard and recursive, with or without bilateral filtering; signals, images, cubes), the
``Coefficients`` operators, ``denoise``, ``enhance``, ``wow``, ``convolution``, ``sdev_loc`` and
``generalized_anscombe``, ``atrous_convolution``, ``richardson_lucy`` (with ``uniform_init`` the
reference itself keeps the estimate in float32, ref utils.py:233, and so does the float32 engine).  1-D signals run as 1 x N images with the 1-D branch's 'mirror' border and (Z, Y, X) cubes
as (Z*Y) x X images (per-slice 2-D filter + axis-0 filter, ref:46-64).  There is deliberately no
CPU fallback.
"""
import copy

import numpy as np

from . import _lib
from ._lib import (PLANE_INPUT, PLANE_NONE, PLANE_OUT, PLANE_SCRATCH, FLAG_FUSED, Plan, Plan64,
                   default_context, acquire_plan, acquire_plan64, release_plan)

__all__ = ['AtrousTransform', 'B3spline', 'Triangle', 'Coefficients', 'generalized_anscombe',
           'convolution']

_NOISE_PLANE = PLANE_SCRATCH(5)   # ndarray noise maps live here (ref:133)
_TMP_PLANE = PLANE_SCRATCH(4)


# ------------------------------------------------------------------------------------------
# scaling functions (ref:152-287): host constants; the tables are data and part of the
# numerical contract (thresholds scale with sigma_e).
# ------------------------------------------------------------------------------------------
class AbstractScalingFunction:
    """Base class: 1-D taps -> separable N-D kernel, dilated kernel, noise tables."""

    coefficients_1d = None
    sigma_e_1d = sigma_e_2d = sigma_e_3d = None
    sigma_e_1d_bilateral = sigma_e_2d_bilateral = sigma_e_3d_bilateral = None
    _family = None   # engine enum; None for user-defined taps

    def __init__(self, name, n_dim):
        self.name = name
        self.n_dim = n_dim
        self.kernel = self.make_kernel()

    @property
    def coefficients_2d(self):
        return np.outer(self.coefficients_1d, self.coefficients_1d)

    @property
    def coefficients_3d(self):
        t = self.coefficients_1d
        return np.einsum('i,j,k->ijk', t, t, t)

    def make_kernel(self):
        by_dim = {1: lambda: self.coefficients_1d, 2: lambda: self.coefficients_2d,
                  3: lambda: self.coefficients_3d}
        if self.n_dim not in by_dim:
            raise ValueError("Unsupported number of dimensions")          # ref:189
        return by_dim[self.n_dim]()

    def atrous_kernel(self, scale):
        """Zero-stuffed ('a trous') kernel at dilation 2**scale (ref:191-197)."""
        step = 2 ** scale
        out = np.zeros([(n - 1) * step + 1 for n in self.kernel.shape])
        out[(slice(None, None, step),) * self.n_dim] = self.kernel
        return out

    def sigma_e(self, bilateral=None):
        """Per-scale std of the coefficients of unit white noise (ref:199-219)."""
        suffix = "" if bilateral is None else "_bilateral"
        if self.n_dim not in (1, 2, 3):
            raise ValueError("Unsupported number of dimensions")          # ref:208,218
        return getattr(self, f"sigma_e_{self.n_dim}d{suffix}")

    def compute_noise_weights(self, n_scales, n_trials=100, bilateral=None):
        """Monte-Carlo calibration of sigma_e (ref:221-229) on top of the GPU transform."""
        transform = AtrousTransform(self.__class__, bilateral=bilateral)
        std = np.zeros(n_scales)
        for _ in range(n_trials):
            size = (len(self.sigma_e_1d) * 2 ** n_scales,) * self.n_dim
            data = np.random.normal(size=size).astype(np.float32)
            coefficients = transform(data, n_scales)
            # per-plane standard deviation from the device-side fp64 moments (wt_reduce): the
            # planes are not downloaded
            plan = coefficients._device()
            npix = float(plan.H) * float(plan.W)
            for s in range(n_scales):
                tot, tot2, _, _ = plan.reduce(s)
                std[s] += np.sqrt(max(tot2 / npix - (tot / npix) ** 2, 0.0))
        return std / n_trials


class Triangle(AbstractScalingFunction):
    """Triangle scaling function, 3 taps (ref:232-258; Starck & Murtagh, appendix A)."""

    _family = _lib.TRIANGLE
    coefficients_1d = np.array([1 / 4, 1 / 2, 1 / 4])
    sigma_e_1d = np.array([0.60840933, 0.33000059, 0.21157957, 0.145824, 0.10158388,
                           0.07155912, 0.04902655, 0.03529812, 0.02409187, 0.01722846,
                           0.01144442])
    sigma_e_2d = np.array([0.7999247, 0.27308452, 0.11998217, 0.05793947, 0.0288104,
                           0.01447795, 0.00733832, 0.0037203, 0.00192882, 0.00098568,
                           0.00048533])
    sigma_e_3d = np.array([0.89736751, 0.19514386, 0.06239262, 0.02311278, 0.00939645])
    sigma_e_2d_bilateral = np.array([0.31063172, 0.34575647, 0.23712331, 0.13559906,
                                     0.07172004, 0.03665405, 0.01850046, 0.00928768,
                                     0.00465967, 0.00234445, 0.00119249])
    sigma_e_3d_bilateral = np.array([0.3828863, 0.36182913, 0.19520299, 0.08498861,
                                     0.03363142])

    def __init__(self, *args, **kwargs):
        super().__init__('triangle', *args, **kwargs)


class B3spline(AbstractScalingFunction):
    """B3-spline scaling function, 5 taps (ref:261-287; Starck & Murtagh, appendix A)."""

    _family = _lib.B3SPLINE
    coefficients_1d = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16])
    sigma_e_1d = np.array([0.72514976, 0.28538683, 0.17901161, 0.12222841, 0.08469601,
                           0.06027006, 0.04242257, 0.02919823, 0.01805671, 0.01383672,
                           0.00943623])
    sigma_e_2d = np.array([8.907e-01, 2.0072e-01, 8.5551e-02, 4.1261e-02, 2.0470e-02,
                           1.0232e-02, 5.1435e-03, 2.6008e-03, 1.3161e-03, 6.7359e-04,
                           4.0040e-04])
    sigma_e_3d = np.array([0.95633954, 0.12491933, 0.03933029, 0.01489642, 0.0064108])
    sigma_e_2d_bilateral = np.array([0.38234752, 0.24305799, 0.16012153, 0.10633541,
                                     0.07083733, 0.04728659, 0.03163678, 0.02122341,
                                     0.01429102, 0.00952376])
    sigma_e_3d_bilateral = np.array([0.44111772, 0.3552894, 0.16137159, 0.05769064,
                                     0.01932497])

    def __init__(self, *args, **kwargs):
        super().__init__('b3spline', *args, **kwargs)


_BUILTIN_TAPS = {_lib.TRIANGLE: np.array([1 / 4, 1 / 2, 1 / 4]),
                 _lib.B3SPLINE: np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16])}


def _f64_source(arr):
    """The array a float64 plan uploads: integer images and big-endian ones (what a FITS file holds)
    stay as they are (Plan64.upload widens / byte-swaps them on the device: a quarter or less of the
    bytes over PCIe and no host astype - the reference's recast, ref:297, 319-320, happens there);
    everything else becomes contiguous float64."""
    a = np.asarray(arr)
    if a.ndim == 2 and Plan64.device_widens(a.dtype):       # integers; big-endian floats (FITS)
        return np.ascontiguousarray(a)
    return np.ascontiguousarray(a, dtype=np.float64)


def _family_of(scaling_function, ndim=2):
    """Engine family of a scaling-function class or instance: the built-in enum, or - for a
    user-defined AbstractScalingFunction subclass (ref:152-229) - the tuple of its 1-D taps,
    which `_lib.Plan` turns into a plan with run-time taps (generic kernels).  The engine
    correlates like cv2.filter2D (ref:39-45); the reference's 1-D branch is a true convolution
    (scipy.ndimage.convolve, ref:65-69), so 1-D plans get the taps reversed (a no-op for
    symmetric taps)."""
    taps = getattr(scaling_function, "coefficients_1d", None)
    if taps is None:
        raise ValueError("scaling function without coefficients_1d")
    taps = np.asarray(taps, dtype=np.float64).ravel()
    fam = getattr(scaling_function, "_family", None)
    if fam is not None and np.array_equal(taps, _BUILTIN_TAPS[fam]):   # not a re-tapped subclass
        return fam
    if taps.size % 2 == 0 or taps.size > 15:
        raise NotImplementedError("user-defined scaling functions need an odd number of taps "
                                  "(at most 15) in the HIP engine")
    return tuple(float(t) for t in (taps[::-1] if ndim == 1 else taps))


def _taps_f64(scaling_function, ndim):
    """1-D taps of a scaling function for the float64 engine, in the engine's correlation order
    (reversed for 1-D signals, whose smoothing is scipy's convolution, ref:65-69)."""
    taps = np.asarray(scaling_function.coefficients_1d, dtype=np.float64).ravel()
    if taps.size % 2 == 0 or taps.size > 15:
        raise NotImplementedError("scaling functions need an odd number of taps (at most 15) "
                                  "in the HIP engine")
    return tuple(float(t) for t in (taps[::-1] if ndim == 1 else taps))


_PAD_MODES = {"symmetric": 0, "reflect": 1, "edge": 2, "wrap": 3, "constant": 4}


def _needs_generic(scaling_function):
    """user-defined scaling functions the separable run-time-tap kernels do not take (an even
    number of taps, or more than 15): served tap by tap through the generic operator"""
    taps = np.asarray(getattr(scaling_function, "coefficients_1d", ()), dtype=np.float64).ravel()
    return taps.size > 0 and (taps.size % 2 == 0 or taps.size > 15)


def _reference_taps(kernel, s):
    """Tap list of the reference's own loop (ref:76-94) for an n-D ``kernel`` at scale ``s``:
    (centre weight, offsets (n, 3) as (dz, dy, dx), weights) in the reference's tap order - the
    tap at kernel index i of an axis of size n reads the sample (n - 1 - i - n // 2) * 2**s away
    (for an even n two taps land on the centre sample and none on one side, as in the reference)."""
    k = np.asarray(kernel, dtype=np.float64)
    hw = tuple(n // 2 for n in k.shape)
    d = 2 ** s
    offs, wts = [], []
    for idx in np.ndindex(*k.shape):
        if idx == hw:
            continue
        offs.append([0] * (3 - k.ndim) + [(n - 1 - i - h) * d for i, n, h in zip(idx, k.shape, hw)])
        wts.append(k[idx])
    return float(k[hw]), np.asarray(offs, dtype=np.int32).reshape(-1, 3), np.asarray(wts, dtype=np.float64)


def _filter_taps(kernel, s, convolve=False):
    """Tap list of convolution() (ref:35-69) for an n-D scaling-function kernel at scale ``s``: the
    zero-stuffed kernel has n = (K - 1) * 2**s + 1 samples per axis and its centre is n // 2 (cv2's
    default anchor, scipy's origin); cv2.filter2D correlates (2-D / 3-D: the sample at
    i * 2**s - n // 2), scipy.ndimage.convolve convolves (1-D: the sample at n // 2 - i * 2**s)."""
    k = np.asarray(kernel, dtype=np.float64)
    d = 2 ** s
    offs, wts = [], []
    for idx in np.ndindex(*k.shape):
        o = [i * d - ((n - 1) * d + 1) // 2 for i, n in zip(idx, k.shape)]
        offs.append([0] * (3 - k.ndim) + ([-v for v in o] if convolve else o))
        wts.append(k[idx])
    return np.asarray(offs, dtype=np.int32).reshape(-1, 3), np.asarray(wts, dtype=np.float64)


def _generic_plan(shape, f64, level=0):
    rows, cols = _plane_shape(shape)
    if f64:
        return acquire_plan64(default_context(), rows, cols, (1.0,), level)
    return acquire_plan(default_context(), rows, cols, _lib.B3SPLINE, level)


_SEP_PLANES = (PLANE_SCRATCH(8), PLANE_SCRATCH(9))      # intermediates of the axis-by-axis generic filter


def _generic_filter(plan, scaling_function, ndim, shape, src, dst, offset_scale, centre_scale, pad_mode, dilation=1):
    """The scaling function's kernel - an outer product of its 1-D taps (ref:170-187) - through the
    generic tap-list operator, AXIS BY AXIS (round 4): K taps per axis instead of K**ndim per sample (17
    taps on an image: 34 instead of 289).  Tap j of an axis reads the sample at j * offset_scale - c,
    c = ((K - 1) * centre_scale + 1) // 2 when the operator is the zero-stuffed kernel of the standard
    algorithm (offset_scale = centre_scale = 2**s: cv2's / scipy's centre n // 2 of the stuffed kernel),
    or (j - K // 2) * offset_scale when it is the BASE kernel applied to a polyphase sub-array (the
    recursive algorithm: centre_scale = 0).  Signals convolve (scipy.ndimage.convolve, ref:65-69): the
    offsets change sign.  Rounding differs from the K**ndim-tap sum in the last bits, as any separable
    evaluation does; two taps gain nothing from the split and keep the single launch."""
    taps = np.asarray(scaling_function.coefficients_1d, dtype=np.float64).ravel()
    K = taps.size
    if centre_scale:
        c = ((K - 1) * centre_scale + 1) // 2
        o = np.arange(K) * offset_scale - c
    else:
        o = (np.arange(K) - K // 2) * offset_scale
    if ndim == 1:
        o = -o
    depth = shape[0] if ndim == 3 else 0
    if ndim == 1 or K < 3:
        offs = np.zeros((K,) * ndim + (3,), dtype=np.int32)
        wts = np.ones((K,) * ndim)
        for ax in range(ndim):
            sh = [1] * ndim
            sh[ax] = K
            offs[..., 3 - ndim + ax] = o.reshape(sh)
            wts = wts * taps.reshape(sh)
        plan.taps_conv(src, PLANE_NONE, dst, offs.reshape(-1, 3), wts.ravel(), None, depth=depth, pad_mode=pad_mode,
                       dilation=dilation)
        return
    cur = src
    for i, ax in enumerate(range(ndim - 1, -1, -1)):              # x first, then y (, then z)
        nxt = dst if i == ndim - 1 else _SEP_PLANES[i]
        # (round 5: tiled kernels - LDS row segments along x, an LDS ring of rows down the polyphase chains along
        #  y / z - instead of K global loads per sample; same arithmetic, identical bits: wt_axis_filter)
        plan.axis_filter(cur, nxt, 3 - ndim + ax, o, taps, depth=depth, pad_mode=pad_mode, dilation=dilation)
        cur = nxt


def _generic_smooth(plan, scaling_function, ndim, shape, src, dst, s):
    """conv_s (ref:35-69) through the generic tap-list operator: 'mirror' border for signals (ref:65-69),
    BORDER_REFLECT = 'symmetric' otherwise (ref:39-63)"""
    _generic_filter(plan, scaling_function, ndim, shape, src, dst, 2 ** s, 2 ** s,
                    _PAD_MODES["reflect" if ndim == 1 else "symmetric"])


def _plane_shape(shape):
    """(rows, cols) of the 2-D image an array of this shape is stored as: 1 x N, H x W, (Z*Y) x X"""
    if len(shape) == 1:
        return 1, shape[0]
    if len(shape) == 3:
        return shape[0] * shape[1], shape[2]
    return tuple(shape)


def _is_f64(arr):
    """the reference would compute this input in float64 (ref:297,319-320)"""
    return _result_dtype(arr) == np.float64


def _is_1d(arr):
    return np.ndim(arr) == 1


def _to_f32_row(arr):
    """1-D signal -> the (1, N) float32 image the engine runs it as."""
    return np.ascontiguousarray(arr, dtype=np.float32).reshape(1, -1)


def _to_f32_image(arr, what="arr"):
    arr = np.asarray(arr)
    if arr.ndim > 3:
        raise ValueError("Unsupported number of dimensions")              # ref:317
    if arr.ndim != 2:
        raise NotImplementedError(
            f"{what}: the HIP engine covers the 2-D path only ({arr.ndim}-D arrays are out of "
            "scope, SURVEY.md section 8)")
    return np.ascontiguousarray(arr, dtype=np.float32)


def _f32_source(arr, what="arr"):
    """The array a float32 plan uploads: integer and byte-swapped images (uint8 pictures, raw big-endian
    FITS integers - what the reference does not recast to float64) stay as they are and are widened on
    the device (Plan.upload); everything else becomes contiguous float32 (_to_f32_image)."""
    a = np.asarray(arr)
    if a.ndim == 2 and _lib.device_widens(a.dtype):
        return np.ascontiguousarray(a)
    return _to_f32_image(arr, what)


_RECAST = [np.dtype(t) for t in (np.int32, np.int64, '>f4', '>f8', 'int16', 'uint16', 'int32', 'uint32')]


def _result_dtype(arr):
    """dtype of what the reference hands back for this input: float64 stays float64 and the
    types of ref:297 are recast to float64 (ref:319-320); everything else is served as float32
    (the engine's compute type)."""
    dt = np.asarray(arr).dtype
    return np.dtype(np.float64) if (dt == np.float64 or dt in _RECAST) else np.dtype(np.float32)


_SOFT_SIG_DTYPE = np.float64 if int(np.__version__.split(".")[0]) >= 2 else np.float32   # NEP 50


# ------------------------------------------------------------------------------------------
# pointwise / single-operator entry points
# ------------------------------------------------------------------------------------------
def generalized_anscombe(signal, alpha=1, g=0, sigma=0, inverse=False):
    """Generalised Anscombe variance-stabilising transform and its algebraic inverse
    (ref:14-21), evaluated on the GPU in float32."""
    arr = np.asarray(signal)
    if arr.ndim > 3:
        raise ValueError("Unsupported number of dimensions")
    shape = arr.shape
    # pointwise: any dimensionality runs as a (rows, last axis) image
    f64 = _is_f64(arr)
    img = np.ascontiguousarray(arr, dtype=np.float64 if f64 else np.float32).reshape(
        -1, shape[-1] if arr.ndim else 1)
    if f64:
        plan = acquire_plan64(default_context(), img.shape[0], img.shape[1], (1.0,), 0)
    else:
        plan = acquire_plan(default_context(), img.shape[0], img.shape[1], _lib.B3SPLINE, 0)
    try:
        plan.upload(PLANE_INPUT, img)
        plan.anscombe(PLANE_INPUT, PLANE_OUT, alpha, g, sigma, inverse)
        return plan.download(PLANE_OUT).reshape(shape).astype(_result_dtype(arr), copy=False)
    finally:
        release_plan(plan)


def convolution(arr, scaling_function, s=0, output=None):
    """Dilated smoothing with ``scaling_function`` at scale ``s``; symmetric borders.
    Mirrors ref:35-45 (2-D branch: cv2.filter2D with the zero-stuffed kernel,
    BORDER_REFLECT).  ``output`` is written in place and returned, as in the reference.
    1-D arrays take the reference's 1-D branch (ref:65-69: scipy 'mirror' border) as a
    1 x N image under the engine's mirror border rule."""
    one_d = _is_1d(arr)
    three_d = np.ndim(arr) == 3
    if _needs_generic(scaling_function) and np.ndim(arr) in (1, 2, 3):
        # an even number of taps or more than 15: tap by tap (wt_taps_conv)
        f64 = _is_f64(arr)
        a = np.ascontiguousarray(arr, dtype=np.float64 if f64 else np.float32)
        plan = _generic_plan(a.shape, f64)
        try:
            plan.upload(PLANE_INPUT, a.reshape(plan.shape))
            _generic_smooth(plan, scaling_function, a.ndim, a.shape, PLANE_INPUT, PLANE_OUT, s)
            res = plan.download(PLANE_OUT).reshape(a.shape).astype(_result_dtype(arr), copy=False)
        finally:
            release_plan(plan)
        if output is None:
            return res
        output[...] = res
        return output
    if _is_f64(arr) and np.ndim(arr) in (1, 2, 3):
        a = _f64_source(arr)
        plan = acquire_plan64(default_context(), *_plane_shape(a.shape),
                              _taps_f64(scaling_function, a.ndim), 0)
        try:
            if one_d:
                plan.set_border(2)
            plan.upload(PLANE_INPUT, a.reshape(plan.shape))
            if three_d:
                plan.smooth3d(PLANE_INPUT, PLANE_OUT, s, a.shape[0])
            else:
                plan.smooth(PLANE_INPUT, PLANE_OUT, s)
            res = plan.download(PLANE_OUT).reshape(a.shape)
        finally:
            release_plan(plan)
        if output is None:
            return res
        output[...] = res
        return output
    if three_d:
        cube = np.ascontiguousarray(arr, dtype=np.float32)
        img = cube.reshape(cube.shape[0] * cube.shape[1], cube.shape[2])
    else:
        img = _to_f32_row(arr) if one_d else _f32_source(arr)
    fam = _family_of(scaling_function, 1 if one_d else 2)
    plan = acquire_plan(default_context(), img.shape[0], img.shape[1], fam, 0)
    try:
        if one_d:
            plan.set_border(2)
        plan.upload(PLANE_INPUT, img)
        if three_d:
            plan.smooth3d(PLANE_INPUT, PLANE_OUT, s, cube.shape[0])
        else:
            plan.smooth(PLANE_INPUT, PLANE_OUT, s)
        res = plan.download(PLANE_OUT)
        res = res.reshape(np.shape(arr)).astype(_result_dtype(arr), copy=False)
    finally:
        release_plan(plan)
    if output is None:
        return res
    output[...] = res
    return output


def sdev_loc(image, scaling_function, s=0, variance=False):
    """Local standard deviation (or variance) at scale ``s`` (ref:24-32)."""
    if _is_f64(image) and np.ndim(image) == 2:
        a = _f64_source(image)
        plan = acquire_plan64(default_context(), a.shape[0], a.shape[1], _taps_f64(scaling_function, 2), 0)
        try:
            plan.upload(PLANE_INPUT, a)
            plan.local_variance(PLANE_INPUT, PLANE_OUT, s, 1.0, 1.0, take_sqrt=not variance)
            return plan.download(PLANE_OUT)
        finally:
            release_plan(plan)
    img = _f32_source(image, "image")
    plan = acquire_plan(default_context(), img.shape[0], img.shape[1],
                        _family_of(scaling_function), 0)
    try:
        plan.upload(PLANE_INPUT, img)
        plan.local_variance(PLANE_INPUT, PLANE_OUT, s, 1.0, 1.0, take_sqrt=not variance)
        return plan.download(PLANE_OUT)
    finally:
        release_plan(plan)


def atrous_convolution(image, kernel, bilateral_variance=None, s=0, mode="symmetric",
                       output=None):
    """Dilated convolution, optionally range-weighted by ``bilateral_variance`` (ref:74-105).

    2-D images with a SEPARABLE kernel (the outer product of an odd number <= 15 of 1-D taps with
    itself - every AbstractScalingFunction kernel, ref:170-173) under ``mode='symmetric'`` run on
    the tuned kernels.  Everything else the reference accepts - any ``kernel`` array of the image's
    dimensionality (non-separable, even-sized, large), signals and cubes, and the np.pad modes
    'symmetric', 'reflect', 'edge', 'wrap' and 'constant' - runs tap by tap on the generic
    operator (wt_taps_conv), in the reference's tap order; the remaining np.pad modes are padded by
    np.pad itself, as in the reference, and filtered by the same operator."""
    kernel = np.asarray(kernel)
    nd = np.ndim(image)
    if nd not in (1, 2, 3) or kernel.ndim != nd:
        raise ValueError("Unsupported number of dimensions")
    if mode not in _PAD_MODES:
        # The other np.pad modes ('linear_ramp', 'maximum', 'mean', 'median', 'minimum', 'empty', or a
        # callable): the reference hands `mode` straight to np.pad (ref:77) - so does this path, with the
        # reference's own pad widths; the padded array then goes through the generic operator (whose
        # taps never leave it for the samples that are kept) and the interior is cropped on the way
        # back.  Border synthesis on the host, the arithmetic on the GPU.  np.pad raises for unknown modes.
        img = np.asarray(image)
        widths = [(n // 2 * 2 ** s,) * 2 for n in kernel.shape]                     # ref:76-77
        padded = np.pad(img, widths, mode=mode)
        var = None
        if bilateral_variance is not None:                   # (used at the centre sample only, ref:97)
            var = np.pad(np.broadcast_to(np.asarray(bilateral_variance, dtype=np.float64), img.shape), widths,
                         mode="constant", constant_values=1.0)
        res = atrous_convolution(padded, kernel, var, s, mode="constant")
        res = res[tuple(slice(w[0], w[0] + n) for w, n in zip(widths, img.shape))]
        if output is None:
            return np.ascontiguousarray(res)
        output[...] = res
        return output
    f64 = _is_f64(image)                                  # float64 engine (ref:319-320)
    fam = None
    flags = 0
    if nd == 2 and mode == "symmetric":
        for cls in (Triangle, B3spline):
            k = cls(2).kernel
            if kernel.shape == k.shape and np.allclose(kernel, k, rtol=1e-6, atol=0):
                fam = cls._family
        if fam is None:
            # kernel = outer(t, t): t = centre row / sqrt(centre).  The reference's tap loop is a true
            # convolution (ref:87-91) and the engine correlates: the plan gets the taps reversed and
            # the range-weighted kernel is told so (wt_bilateral_conv flag bit 3).
            k64 = np.asarray(kernel, dtype=np.float64)
            ok = k64.shape[0] == k64.shape[1] and k64.shape[0] % 2 == 1 \
                and k64.shape[0] <= 15 and k64[k64.shape[0] // 2, k64.shape[0] // 2] > 0
            if ok:
                hw = k64.shape[0] // 2
                taps = k64[hw] / np.sqrt(k64[hw, hw])
                ok = np.allclose(np.multiply.outer(taps, taps), k64, rtol=1e-6,
                                 atol=1e-7 * np.abs(k64).max())
            if ok:
                fam = tuple(float(t) for t in taps[::-1])
                flags = _lib.FLAG_TAPS_REVERSED
    if fam is None:
        # the general case: the reference's loop, tap by tap
        a = np.ascontiguousarray(image, dtype=np.float64 if f64 else np.float32)
        kc, offs, wts = _reference_taps(kernel, s)
        plan = _generic_plan(a.shape, f64)
        try:
            plan.upload(PLANE_INPUT, a.reshape(plan.shape))
            var_plane = PLANE_NONE
            if bilateral_variance is not None:
                var = np.broadcast_to(np.asarray(bilateral_variance, a.dtype), a.shape)
                plan.upload(_TMP_PLANE, np.ascontiguousarray(var).reshape(plan.shape))
                var_plane = _TMP_PLANE
            plan.taps_conv(PLANE_INPUT, var_plane, PLANE_OUT, offs, wts, kc,
                           depth=a.shape[0] if nd == 3 else 0, pad_mode=_PAD_MODES[mode])
            res = plan.download(PLANE_OUT).reshape(a.shape)
        finally:
            release_plan(plan)
        if output is None:
            return res
        output[...] = res
        return output
    img = _f64_source(image) if f64 else _f32_source(image, "image")
    if f64:
        if not isinstance(fam, tuple):                   # built-in family: its (symmetric) taps
            fam = _taps_f64(Triangle if fam == _lib.TRIANGLE else B3spline, 2)
        plan = acquire_plan64(default_context(), img.shape[0], img.shape[1], fam, 0)
    else:
        plan = acquire_plan(default_context(), img.shape[0], img.shape[1], fam, 0)
    try:
        plan.upload(PLANE_INPUT, img)
        if bilateral_variance is None:
            plan.smooth(PLANE_INPUT, PLANE_OUT, s)
        else:
            var = np.broadcast_to(np.asarray(bilateral_variance, np.float64 if f64 else np.float32), img.shape)
            plan.upload(_TMP_PLANE, var)
            plan.bilateral_conv(PLANE_INPUT, _TMP_PLANE, PLANE_OUT, s, flags)
        res = plan.download(PLANE_OUT)
    finally:
        release_plan(plan)
    if output is None:
        return res
    output[...] = res
    return output


# ------------------------------------------------------------------------------------------
# Coefficients (ref:108-149) - device-resident planes with a lazily materialised host mirror
# ------------------------------------------------------------------------------------------
class Coefficients:
    """``level+1`` coefficient planes: 0..level-1 detail, ``level`` the final smooth.

    Planes live in HBM (float32 planes of a Plan, or float64 planes of a Plan64 for float64 /
    promoted inputs).  ``.data`` materialises an ndarray mirror of that dtype on first access;
    from then on the mirror is treated as user-owned (the reference's idiom is in-place
    edits such as ``coefficients.data[s] *= ...``): every device operation first re-uploads
    it and afterwards refreshes it in place, so references held by the caller stay valid.
    """

    def __init__(self, data, scaling_function, bilateral=None, _shape=None, _dtype=None):
        self.scaling_function = scaling_function
        self.bilateral = bilateral
        self.noise = None
        self._plan = None
        self._host = None
        self._noise_uploaded = None
        self._sum_ok = False        # PLANE_OUT holds np.sum(planes, axis=0) of the CURRENT planes (_sum_valid)
        self._host_sum = None       # ... and this host array too (with_sum=True transforms of host images)
        # logical shape of one plane: (N,), (H, W) or (Z, Y, X); the engine stores it as a 2-D
        # image: 1 x N, H x W or (Z*Y) x X
        if isinstance(data, (Plan, Plan64)):
            self._plan = data
            self._nplanes = data.max_level + 1
            if _shape is not None:
                self._shape = tuple(_shape)
            elif getattr(scaling_function, "n_dim", 2) == 1:
                self._shape = (data.shape[1],)
            else:
                self._shape = tuple(data.shape)
        else:
            data = np.asarray(data)
            if data.ndim not in (2, 3, 4):
                raise ValueError("Unsupported number of dimensions")
            if _dtype is None:
                _dtype = _result_dtype(data)
            self._host = np.ascontiguousarray(data, dtype=_dtype)
            self._nplanes = self._host.shape[0]
            self._shape = tuple(self._host.shape[1:])
        # dtype of the host mirror and of every array handed back: what the reference would return
        # for the transform's input.  float64 objects compute on the float64 engine (Plan64)
        self._dtype = np.dtype(np.float32 if _dtype is None else _dtype)
        if isinstance(data, Plan64):
            self._dtype = np.dtype(np.float64)
        self._ndim = len(self._shape)

    def __del__(self):
        try:
            release_plan(self._plan)     # planes go back to the pool
        except Exception:
            pass

    # -- copies / pickling: a copy is a host-backed object with its own planes ---------------
    def __deepcopy__(self, memo):
        other = Coefficients(np.array(self.data, copy=True), copy.deepcopy(self.scaling_function, memo),
                             copy.deepcopy(self.bilateral, memo))
        other.noise = copy.deepcopy(self.noise, memo)
        return other

    def __copy__(self):
        return self.__deepcopy__({})

    def __reduce__(self):
        return (_rebuild_coefficients, (np.array(self.data, copy=True), self.scaling_function,
                                        self.bilateral, self.noise))

    # -- host mirror -------------------------------------------------------------------
    def _img_shape(self):
        return self._shape

    def _plane_hw(self):
        """(rows, cols) of the 2-D image a plane is stored as"""
        if self._ndim == 1:
            return 1, self._shape[0]
        if self._ndim == 3:
            return self._shape[0] * self._shape[1], self._shape[2]
        return self._shape

    def _as_plane(self, a):
        """host plane in the engine's 2-D layout (1 x N, H x W, (Z*Y) x X)"""
        return a.reshape(self._plane_hw())

    def _from_plane(self, a):
        return a.reshape(self._shape)

    @property
    def data(self):
        if self._host is None:
            host = _lib.host_empty((self._nplanes,) + self._img_shape(), self._plan.ctx, self._dtype)
            self._host = host
            self._refresh_host(range(self._nplanes))
        return self._host

    @data.setter
    def data(self, value):
        value = np.ascontiguousarray(value, dtype=self._dtype)
        if value.ndim != self._ndim + 1 or value.shape[0] != self._nplanes:
            raise ValueError("Coefficients.data must keep its (level+1, ...) shape")
        self._host = value

    def _device(self):
        """Entry point of every device operation: the plan with planes up to date
        (re-uploading a user-owned mirror).  Call ONCE per public operation; helpers that
        run inside an operation use ``self._plan`` directly."""
        if self._plan is None:
            H, W = self._plane_hw()
            if _needs_generic(self.scaling_function):
                # scaling functions with an even number of taps or more than 15 run on the generic
                # tap-list operator: their planes live on a storage-only plan (the plan's own taps are
                # never used), which is what a copy / an unpickled / a hand-built object needs back
                self._plan = _generic_plan(self._shape, self._dtype == np.float64, self._nplanes - 1)
            elif self._dtype == np.float64:
                self._plan = acquire_plan64(default_context(), H, W,
                                            _taps_f64(self.scaling_function, self._ndim), self._nplanes - 1)
            else:
                self._plan = acquire_plan(default_context(), H, W,
                                          _family_of(self.scaling_function), self._nplanes - 1)
        if self._host is not None:
            for s in range(self._nplanes):
                self._plan.upload(s, self._as_plane(self._host[s]))
            self._sum_valid = False          # a user-owned mirror may have been edited
        return self._plan

    def _refresh_host(self, planes):
        if self._host is not None:
            plan_dtype = np.float64 if isinstance(self._plan, Plan64) else np.float32
            for s in planes:
                if self._host.dtype == plan_dtype:
                    self._plan.download(s, self._as_plane(self._host[s]))
                else:                        # float64 mirror of float32 planes
                    self._as_plane(self._host[s])[...] = self._plan.download(s)

    @property
    def _sum_valid(self):
        return self._sum_ok

    @_sum_valid.setter
    def _sum_valid(self, value):
        # whatever (re)defines the content of PLANE_OUT invalidates the host copy of the synthesis
        self._sum_ok = bool(value)
        self._host_sum = None

    # -- reference interface -----------------------------------------------------------
    def __len__(self):
        return self._nplanes                                              # ref:116

    def __array__(self, dtype=None, copy=None):
        d = self.data                                                     # ref:119
        return d if dtype is None else d.astype(dtype, copy=False)

    @property
    def shape(self):
        return (self._nplanes,) + tuple(self._img_shape())

    @property
    def sigma_e(self):
        return self.scaling_function.sigma_e(bilateral=self.bilateral)    # ref:122-124

    def get_noise(self):
        """MAD noise estimate: median(|w_0|) / 0.6745 / sigma_e[0] (ref:126-127).  The exact
        median is a radix select on the GPU; the scalar divisions follow numpy's promotion."""
        self._device()
        return self._noise_from_device()

    def _noise_from_device(self):
        return self._plan.abs_median(0) / 0.6745 / self.sigma_e[0]

    def _tau(self, sigma, scale, soft=True):
        """(tau, noise_plane) or None when the significance is identically one
        (sigma == 0, ref:142-143; scalar noise == 0, ref:133-135).  Runs inside a device
        operation (planes already in sync)."""
        if sigma == 0:
            return None
        if self.noise is None:
            self.noise = self._noise_from_device()                        # ref:131-132 (lazy)
        if type(self.noise) is not np.ndarray:
            if self.noise == 0:
                return None
            tau = float(sigma * self.noise * self.sigma_e[scale])
            if tau < 0:
                # ref:137-141 with a negative threshold: erf(|w / tau|) is erf(|w| / |tau|), and
                # |w| > tau is always true (significance one)
                return (-tau, PLANE_NONE) if soft else None
            return tau, PLANE_NONE
        plan = self._plan
        if self._noise_uploaded is not self.noise:
            plan.upload(_NOISE_PLANE, np.broadcast_to(
                self._as_plane(np.asarray(self.noise, np.float64 if isinstance(plan, Plan64) else np.float32)),
                plan.shape))
            self._noise_uploaded = self.noise
        tau = float(sigma * self.sigma_e[scale])
        if tau < 0 and not soft:
            return None
        return abs(tau), _NOISE_PLANE

    def significance(self, sigma, scale, soft_threshold=True):
        """erf(|w|/tau) (soft) or |w| > tau (hard, bool), tau = sigma*noise*sigma_e[scale]
        (ref:129-143).  Returns a host ndarray like the reference."""
        plan = self._device()
        t = self._tau(sigma, scale, soft_threshold)
        if t is None:
            return np.ones(self._img_shape(), self._dtype)                # ones_like(data[0]), ref:135,143
        plan.significance(scale, _TMP_PLANE, t[0], soft_threshold, t[1])
        sig = self._from_plane(plan.download(_TMP_PLANE))
        # ref:137-141: bool for the hard threshold; the soft one divides by numpy float64 scalars,
        # which under NumPy 2 (NEP 50) makes the ratio - and erf of it - float64 for any data
        return sig.astype(_SOFT_SIG_DTYPE if self._dtype == np.float32 else self._dtype, copy=False) \
            if soft_threshold else sig.astype(bool)

    def denoise(self, sigma, weights=None, soft_threshold=True):
        """In-place ``w_s *= weights[s] * significance(sigma[s], s)`` for the first
        ``len(sigma)`` planes (ref:145-149)."""
        if weights is None:
            weights = (1,) * len(sigma)
        plan = self._device()
        self._sum_valid = False
        touched = []
        for scl, (_, sig, wgt) in enumerate(zip(range(self._nplanes), sigma, weights)):
            t = self._tau(sig, scl, soft_threshold)
            if t is None:
                if wgt != 1:
                    plan.wow_update(scl, PLANE_NONE, 0.0, True, PLANE_NONE, wgt, PLANE_NONE)
                    touched.append(scl)
                continue
            plan.denoise(scl, t[0], wgt, soft_threshold, t[1])
            touched.append(scl)
        self._refresh_host(touched)

    def _lazy_noise_after_rescale(self, sigma, weights):
        """The reference estimates the noise lazily, at the first non-zero threshold (ref:131-132),
        from plane 0 AS IT IS THEN: with sigma[0] == 0 and weights[0] != 1 that plane has already
        been rescaled (ref:149).  The fused forms take all thresholds first, so they leave this
        corner to the plane-by-plane ``denoise``."""
        pairs = list(zip(range(self._nplanes), sigma, weights))
        return self.noise is None and len(pairs) > 1 and pairs[0][1] == 0 and pairs[0][2] != 1 \
            and any(sig != 0 for _, sig, _ in pairs[1:])

    def _denoise_sum(self, sigma, weights=None, soft_threshold=True, write_back=True):
        """denoise(sigma, weights) fused with the plane sum: one pass over the planes
        (wt_denoise_sum).  Same lazy-noise / truncation rules as ``denoise``."""
        if weights is None:
            weights = (1,) * len(sigma)
        if self._lazy_noise_after_rescale(sigma, weights):
            self.denoise(sigma, weights, soft_threshold)                  # the reference's order
            plan = self._plan
            plan.plane_sum(0, self._nplanes, PLANE_OUT)
            self._sum_valid = self._host is None
            return plan
        plan = self._device()
        self._sum_valid = False              # PLANE_OUT becomes the DENOISED sum
        taus, wgts, noise_plane = [], [], PLANE_NONE
        for scl, (_, sig, wgt) in enumerate(zip(range(self._nplanes), sigma, weights)):
            t = self._tau(sig, scl, soft_threshold)
            taus.append(0.0 if t is None else t[0])
            wgts.append(wgt)
            if t is not None and t[1] != PLANE_NONE:
                noise_plane = t[1]
        plan.denoise_sum(self._nplanes, taus, wgts, soft_threshold, noise_plane, write_back)
        if write_back:
            self._refresh_host(range(len(taus)))
        return plan

    # -- numpy reduction hook ------------------------------------------------------------
    def sum(self, axis=None, dtype=None, out=None, **kwargs):
        """``np.sum(coefficients, axis=0)`` (ref utils.py:98,205; README) dispatches here:
        the plane sum runs on the GPU in plane order (bit-identical to numpy's float32
        reduction over axis 0).  Any other reduction falls back to numpy on the mirror."""
        if axis == 0 and dtype is None and not kwargs:
            plan = self._device()
            if not self._sum_valid:           # with_sum=True transforms carried it along already
                plan.plane_sum(0, self._nplanes, PLANE_OUT)
                self._sum_valid = self._host is None
            if self._sum_valid and self._host_sum is not None:
                host, self._host_sum = self._host_sum, None      # came down with the transform
            else:
                host = plan.download(PLANE_OUT)
            res = self._from_plane(host).astype(self._dtype, copy=False)
            if out is None:
                return res
            out[...] = res
            return out
        return np.sum(self.data, axis=axis, dtype=dtype, out=out, **kwargs)


def _decompose_denoise_sum(transform, plan, level, coefficients, sigma, weights=None,
                           soft_threshold=True, write_back=False):
    """Transform (ref:95 / 408-444), ``Coefficients.denoise(sigma, weights)`` (ref:145-149) and
    ``np.sum(coefficients, axis=0)`` (utils.py:98) of the image in ``plan``'s input plane, with the
    threshold step placed BETWEEN the fused passes: the passes that produce the planes to be
    thresholded run first, the MAD noise estimate reads plane 0 (final by then, ref:131-132), one
    kernel thresholds those planes and starts the sum with them (wt_denoise_sum), and the remaining
    passes carry the sum along (wt_decompose_pass_sum) instead of the sum re-reading every plane.
    Same operations in the same order as transform -> denoise -> sum: identical bits.  Falls back
    to exactly that sequence when the schedule is not all fused passes or every plane is
    thresholded.  Result in PLANE_OUT; ``write_back`` also stores the thresholded planes."""
    if weights is None:
        weights = (1,) * len(sigma)
    entries = list(zip(range(level + 1), sigma, weights))                 # zip truncation, ref:148
    n_den = max([scl + 1 for scl, sig, wgt in entries if sig != 0 or wgt != 1], default=0)
    # (float64 plans: the fused passes serve the built-in families' taps, wt64_decompose_pass)
    fam = plan.fused_family if isinstance(plan, Plan64) else (None if plan.custom else plan.family)
    sched = _lib.schedule(fam, level, True) if fam is not None else []
    k, covered = 0, 0
    if transform.bilateral is None and fam is not None and level > 0 and plan.fused_ok(level):
        while k < len(sched) and (covered < n_den or k == 0):
            covered += sched[k][1]
            k += 1
    if k == 0 or k == len(sched) or coefficients._lazy_noise_after_rescale(sigma, weights):
        transform._run(plan, level)
        coefficients._denoise_sum(sigma, weights, soft_threshold, write_back)
        return plan
    coefficients._sum_valid = False
    cur = PLANE_INPUT
    # the MAD noise estimate (lazy in the reference, ref:131-132) will be needed iff some threshold
    # is non-zero and no noise was given: then the first pass histograms |w_0| as it produces it
    # and the estimate is taken right behind it (plane 0 is final by then)
    want_noise = coefficients.noise is None and any(sig != 0 for _, sig, _ in entries[:covered])
    for i in range(k):
        nxt = PLANE_SCRATCH(i & 1)
        first = i == 0 and want_noise
        plan.decompose_pass(cur, nxt, sched[i][0], sched[i][1],
                            FLAG_FUSED | (_lib.FLAG_MEDIAN_HIST if first else 0))
        if first:
            coefficients.noise = coefficients._noise_from_device()
        cur = nxt
    taus, wgts, noise_plane = [], [], PLANE_NONE
    for scl, sig, wgt in entries[:covered]:
        t = coefficients._tau(sig, scl, soft_threshold)
        taus.append(0.0 if t is None else t[0])
        wgts.append(wgt)
        if t is not None and t[1] != PLANE_NONE:
            noise_plane = t[1]
    plan.denoise_sum(covered, taus, wgts, soft_threshold, noise_plane, write_back)
    for i in range(k, len(sched)):
        last = i == len(sched) - 1
        nxt = level if last else PLANE_SCRATCH(i & 1)
        plan.decompose_pass_sum(cur, nxt, sched[i][0], sched[i][1], FLAG_FUSED, PLANE_OUT,
                                first=False, last=last)
        cur = nxt
    coefficients._sum_valid = bool(write_back) and coefficients._host is None
    return plan


def _rebuild_coefficients(data, scaling_function, bilateral, noise):
    c = Coefficients(data, scaling_function, bilateral)
    c.noise = noise
    return c


# ------------------------------------------------------------------------------------------
# AtrousTransform (ref:290-328, 408-444)
# ------------------------------------------------------------------------------------------
class AtrousTransform:
    """Dyadic 'a trous' (stationary) wavelet transform, Starck & Murtagh appendix A."""

    def __init__(self, scaling_function_class=B3spline, bilateral=None, bilateral_scaling=False):
        self.scaling_function_class = scaling_function_class
        self.bilateral = bilateral
        self.bilateral_scaling = bilateral_scaling

    def __call__(self, arr, level, recursive=False, *, with_sum=False, _f64=True):
        """``level`` scales -> ``Coefficients`` with ``level + 1`` planes (ref:307-328).

        ``with_sum=True`` (keyword-only, not in the reference) asks the transform to carry the
        synthesis ``np.sum(coefficients, axis=0)`` through its fused passes (wt_decompose_sum: the
        planes are written as usual but not re-read).  The sum is kept on the device next to the
        planes and handed out by ``np.sum(coefficients, axis=0)`` / ``coefficients.sum(axis=0)``
        as long as no plane has been modified since; it is bit-identical to summing afterwards.
        Without the keyword nothing extra is computed.

        float64 / int inputs (computed in float64 by the reference, ref:297,319-320) run on the
        float64 engine."""
        if np.ndim(arr) in (1, 2, 3) and _needs_generic(self.scaling_function_class):
            return self._call_generic(arr, level, recursive)
        if _f64 and _is_f64(arr) and np.ndim(arr) in (1, 2, 3):
            if recursive:
                return self._recursive(np.asarray(arr), level, self.scaling_function_class(np.ndim(arr)),
                                       np.float64)
            return self._call_f64(arr, level, with_sum)
        if _is_1d(arr):
            return self._call_1d(arr, level, recursive)
        if np.ndim(arr) == 3:
            return self._call_3d(arr, level, recursive)
        img = _to_f32_image(arr) if recursive else _f32_source(arr)
        scaling_function = self.scaling_function_class(img.ndim)
        if recursive:
            return self._recursive(img, level, scaling_function, _result_dtype(arr))
        plan = acquire_plan(default_context(), img.shape[0], img.shape[1],
                            _family_of(scaling_function), level)
        # (the pipelined host call takes float32 rows; an image widened on the device is uploaded whole)
        summed = bool(with_sum) and self.bilateral is None and not plan.custom and img.dtype == np.float32
        host_sum = None
        if summed:
            # ref:432,442 + utils.py:98; the upload, the passes and the download of the synthesis are
            # pipelined over blocks of rows (wt_decompose_sum_host: about one PCIe leg instead of two)
            host_sum = plan.decompose_sum_host(img, level, PLANE_OUT)
        else:
            plan.upload(PLANE_INPUT, img)
            self._run(plan, level)
        coefficients = Coefficients(plan, scaling_function, self.bilateral, _dtype=_result_dtype(arr))
        coefficients._sum_valid = summed
        coefficients._host_sum = host_sum        # handed out once by np.sum(coefficients, axis=0)
        return coefficients

    def _call_generic(self, arr, level, recursive):
        """The transform for a user-defined scaling function with an even number of taps or more than
        15, tap by tap through the generic operator - float32 or float64 planes as the input's type
        asks; signals, images and cubes.

        * standard algorithm (ref:408-444): conv_s = convolution()'s zero-stuffed kernel, centre n // 2
          (_generic_filter: axis by axis); with bilateral filtering (ref:433-440) the variance of sdev_loc from
          conv_s(I) and conv_s(I^2) (ref:24-32), then the reference's own tap loop, range-weighted
          (_reference_taps, ref:74-105), under the symmetric pad.
        * recursive algorithm (ref:330-406): the array is padded once by (n // 2) * 2**(level-1) per
          axis (ref:394-395); at scale s every polyphase sub-array of stride d = 2**s is filtered on
          its own by the BASE operator (ref:354-390) - i.e. the base tap list with offsets times d
          under the polyphase border rule (an out-of-range index is extended inside its own residue
          class: wt_taps_conv_ex pad modes 5 / 6) - and the planes are cropped (ref:405-406).  For an
          even tap count this is NOT the standard algorithm shifted: the base kernel's anchor n // 2
          scales with d, the zero-stuffed kernel's centre does not."""
        f64 = _is_f64(arr)
        a = np.ascontiguousarray(arr, dtype=np.float64 if f64 else np.float32)
        nd = a.ndim
        scaling_function = self.scaling_function_class(nd)
        kernel = np.asarray(scaling_function.kernel, dtype=np.float64)
        sb = None if self.bilateral is None else self._sigma_bilateral(level)
        if recursive:
            if level < 1:
                raise ValueError("recursive=True needs level >= 1")
            pads = [(n // 2) * 2 ** (level - 1) for n in kernel.shape]                # ref:394
            work = np.pad(a, [(w, w) for w in pads], mode="symmetric")               # ref:395
        else:
            pads, work = None, a
        plan = _generic_plan(work.shape, f64, level)
        depth = work.shape[0] if nd == 3 else 0
        plan.upload(PLANE_INPUT, work.reshape(plan.shape))
        if recursive:
            # the base operators' tap lists (scale 0); offsets are multiplied by d below
            r_kc, r_offs, r_wts = _reference_taps(kernel, 0)                         # atrous_convolution on a sub-array
            conv_pad = _lib.PAD_POLY_MIRROR if nd == 1 else _lib.PAD_POLY_SYMMETRIC  # ref:65-69 / :39-63 per sub-array
        sq, mean, var = PLANE_SCRATCH(6), PLANE_SCRATCH(7), _TMP_PLANE     # (scratch 0/1: the smooth planes; 5: noise maps)
        cur = PLANE_INPUT
        for s in range(level):
            nxt = level if s == level - 1 else PLANE_SCRATCH(s & 1)
            d = 2 ** s

            def smooth(src, dst):
                if recursive:         # the base kernel on every sub-array of stride d (ref:371-372)
                    _generic_filter(plan, scaling_function, nd, work.shape, src, dst, d, 0, conv_pad, dilation=d)
                else:
                    _generic_smooth(plan, scaling_function, nd, work.shape, src, dst, s)       # ref:432
            if sb is None:
                smooth(cur, nxt)
            else:
                # ref:434-436 / 375-377: variance = sdev_loc(c, sf, s, variance=True) * sigma_b[s]**2 (* (s+1))
                plan.binary("mul", cur, cur, sq)
                smooth(sq, var)
                smooth(cur, mean)
                plan.variance_from_moments(mean, var, var, float(sb[s]) ** 2,
                                           float(s + 1) if self.bilateral_scaling else 1.0)
                if recursive:                                                       # ref:378, on every sub-array
                    plan.taps_conv(cur, var, nxt, r_offs * d, r_wts, r_kc, depth=depth,
                                   pad_mode=_lib.PAD_POLY_SYMMETRIC, dilation=d)
                else:                                                               # ref:439-440
                    kc, offs, wts = _reference_taps(kernel, s)
                    plan.taps_conv(cur, var, nxt, offs, wts, kc, depth=depth, pad_mode=_PAD_MODES["symmetric"])
            plan.binary("sub", cur, nxt, s)                                        # ref:442 / 402-403
            cur = nxt
        if level == 0:
            plan.decompose(PLANE_INPUT, 0)
        if not recursive:
            return Coefficients(plan, scaling_function, self.bilateral, _shape=a.shape if nd == 3 else None,
                                _dtype=_result_dtype(arr))
        out = _generic_plan(a.shape, f64, level)                                     # ref:405-406: remove the pads
        try:
            for s in range(level + 1):
                if nd == 3:
                    Z, Y, X = a.shape
                    Yp = work.shape[1]
                    for z in range(Z):
                        out.copy_window_from(plan, s, s, (z + pads[0]) * Yp + pads[1], pads[2], z * Y, 0, Y, X)
                elif nd == 2:
                    out.copy_window_from(plan, s, s, pads[0], pads[1], 0, 0, a.shape[0], a.shape[1])
                else:
                    out.copy_window_from(plan, s, s, 0, pads[0], 0, 0, 1, a.shape[0])
        finally:
            release_plan(plan)
        return Coefficients(out, scaling_function, self.bilateral, _shape=a.shape if nd == 3 else None,
                            _dtype=_result_dtype(arr))

    def _call_f64(self, arr, level, with_sum=False):
        """Standard algorithm in float64 (ref:408-444 on float64 / promoted input, ref:319-320):
        double planes on a Plan64.  Images with a built-in family run the fused multi-scale passes
        instantiated for double (wt64_decompose decides; with_sum carries the plane sum through
        them); otherwise one generic pass per scale: signals as 1 x N images under the 'mirror'
        border of the 1-D branch (ref:65-69), cubes as (Z*Y) x X images (ref:46-63); with
        bilateral filtering the per-scale sequence of ref:433-442."""
        a = _f64_source(arr)
        nd = a.ndim
        scaling_function = self.scaling_function_class(nd)
        plan = acquire_plan64(default_context(), *_plane_shape(a.shape), _taps_f64(scaling_function, nd), level)
        plan.upload(PLANE_INPUT, a.reshape(plan.shape))
        if self.bilateral is None:
            if nd == 1:
                plan.set_border(2)
            summed = False
            if nd == 3:
                plan.decompose3d(PLANE_INPUT, level, a.shape[0])
            elif nd == 2 and with_sum and level > 0:
                summed = plan.decompose_sum(PLANE_INPUT, level, PLANE_OUT) or True   # (two-step form: the sum is there too)
            else:
                # (images: the first fused pass also histograms |w_0| for a get_noise() that may follow)
                plan.decompose(PLANE_INPUT, level, _lib.FLAG_MEDIAN_HIST if nd == 2 else 0)
            plan.set_border(0)
            c = Coefficients(plan, scaling_function, self.bilateral,
                             _shape=a.shape if nd == 3 else None, _dtype=np.float64)
            c._sum_valid = summed
            return c
        elif nd == 2 and level > 0:
            # images: the per-scale sequence of ref:433-442 in one call (built-in taps: one marching
            # kernel per scale, wt64_decompose_bilateral)
            plan.decompose_bilateral(PLANE_INPUT, level, self._sigma_bilateral(level), self.bilateral_scaling)
        else:
            sb = self._sigma_bilateral(level)
            cur = PLANE_INPUT
            for s in range(level):
                nxt = level if s == level - 1 else PLANE_SCRATCH(s & 1)
                f1 = float(sb[s]) ** 2
                f2 = float(s + 1) if self.bilateral_scaling else 1.0
                if nd == 3:
                    plan.local_variance3d(cur, _TMP_PLANE, s, a.shape[0], f1, f2)
                    plan.bilateral3d_conv(cur, _TMP_PLANE, nxt, s, a.shape[0])
                else:
                    plan.set_border(2 if nd == 1 else 0)           # variance: convolution()'s border
                    plan.local_variance(cur, _TMP_PLANE, s, f1, f2)
                    plan.set_border(0)                             # ref:77: symmetric pad
                    plan.bilateral_conv(cur, _TMP_PLANE, nxt, s,
                                        _lib.FLAG_TAPS_REVERSED if nd == 1 else 0)
                plan.binary("sub", cur, nxt, s)                    # ref:442
                cur = nxt
            if level == 0:
                plan.decompose(PLANE_INPUT, 0)
        return Coefficients(plan, scaling_function, self.bilateral,
                            _shape=a.shape if nd == 3 else None, _dtype=np.float64)

    # the reference's two algorithm entry points return the stacked planes as an ndarray
    # (ref:330-406, 408-444); kept for code that calls them directly
    _recasting_types = [np.int32, np.int64, '>f4', '>f8', 'int16', 'uint16', 'int32', 'uint32']

    def _as_class(self, scaling_function):
        other = AtrousTransform(type(scaling_function), self.bilateral, self.bilateral_scaling)
        return other

    def atrous_standard(self, arr, level, scaling_function):
        """(level + 1, ...) ndarray of planes, standard algorithm (ref:408-444)."""
        return self._as_class(scaling_function)(arr, level, recursive=False).data

    def atrous_recursive(self, arr, level, scaling_function):
        """(level + 1, ...) ndarray of planes, recursive algorithm (ref:330-406)."""
        return self._as_class(scaling_function)(arr, level, recursive=True).data

    def _call_1d(self, arr, level, recursive):
        """1-D signals (ref:65-69, 'mirror' border): run as a 1 x N image with the engine's
        mirror border rule and the per-scale kernels."""
        if recursive:
            return self._recursive(np.asarray(arr), level, self.scaling_function_class(1),
                                   _result_dtype(arr))
        row = _to_f32_row(arr)
        scaling_function = self.scaling_function_class(1)
        plan = acquire_plan(default_context(), 1, row.shape[1], _family_of(scaling_function, 1), level)
        plan.upload(PLANE_INPUT, row)
        if self.bilateral is None:
            plan.set_border(2)
            plan.decompose(PLANE_INPUT, level, 0)
            plan.set_border(0)
            return Coefficients(plan, scaling_function, None, _dtype=_result_dtype(arr))
        # Bilateral (ref:433-440 on a 1-D signal): the variance comes from convolution()'s 1-D
        # branch ('mirror' border, ref:24-32 over :65-69), the range-weighted convolution pads
        # symmetrically (ref:77).  On a 1 x N image the 2-D bilateral kernel reduces to the 1-D
        # one: the taps of the other axis all reflect onto the same row, carry weight e = 1 and
        # factor out of numerator and denominator.
        # (user-defined taps: the plan of a 1-D signal holds them reversed, see _family_of)
        sb = self._sigma_bilateral(level)
        cur = PLANE_INPUT
        for s in range(level):
            nxt = level if s == level - 1 else PLANE_SCRATCH(s & 1)
            plan.set_border(2)
            plan.local_variance(cur, _TMP_PLANE, s, float(sb[s]) ** 2,
                                float(s + 1) if self.bilateral_scaling else 1.0)
            plan.set_border(0)
            plan.bilateral_conv(cur, _TMP_PLANE, nxt, s, _lib.FLAG_TAPS_REVERSED)
            plan.binary("sub", cur, nxt, s)                                # ref:442
            cur = nxt
        if level == 0:
            plan.copy(PLANE_INPUT, 0)
        return Coefficients(plan, scaling_function, self.bilateral, _dtype=_result_dtype(arr))

    def _call_3d(self, arr, level, recursive):
        """(Z, Y, X) cubes (ref:46-64): per-slice 2-D filter then the same filter along axis 0;
        the cube lives on the GPU as a (Z*Y) x X image."""
        if recursive:
            return self._recursive(np.asarray(arr), level, self.scaling_function_class(3),
                                   _result_dtype(arr))
        cube = np.ascontiguousarray(arr, dtype=np.float32)
        Z, Y, X = cube.shape
        scaling_function = self.scaling_function_class(3)
        fam = _family_of(scaling_function)
        plan = acquire_plan(default_context(), Z * Y, X, fam, level)
        plan.upload(PLANE_INPUT, cube.reshape(Z * Y, X))
        if self.bilateral is None:
            plan.decompose3d(PLANE_INPUT, level, Z)
            return Coefficients(plan, scaling_function, None, _shape=(Z, Y, X), _dtype=_result_dtype(arr))
        # bilateral (ref:433-440 on a cube): 3-D variance, then the K^3 range-weighted kernel
        sb = self._sigma_bilateral(level)
        cur = PLANE_INPUT
        for s in range(level):
            nxt = level if s == level - 1 else PLANE_SCRATCH(s & 1)
            plan.local_variance3d(cur, _TMP_PLANE, s, Z, float(sb[s]) ** 2,
                                  float(s + 1) if self.bilateral_scaling else 1.0)
            plan.bilateral3d_conv(cur, _TMP_PLANE, nxt, s, Z)
            plan.binary("sub", cur, nxt, s)                                # ref:442
            cur = nxt
        if level == 0:
            plan.copy(PLANE_INPUT, 0)
        return Coefficients(plan, scaling_function, self.bilateral, _shape=(Z, Y, X), _dtype=_result_dtype(arr))

    def _recursive(self, arr, level, scaling_function, dtype=np.float32):
        """The reference's recursive algorithm (ref:330-406) on the GPU, for signals, images and
        cubes, with or without bilateral filtering.  It pads once by hw*2**(level-1) on every axis
        (ref:394-395), filters every polyphase sub-array on its own with the base operator of its
        dimensionality (ref:354-390) and crops (ref:405-406) - which differs from the standard
        algorithm near the borders from scale 3 on.  Here: the padded array is transformed by the
        per-scale kernels under the plan's POLYPHASE border rule (an out-of-range index reflects
        inside its own residue class modulo the dilation: wt_plan_set_border 1, or 3 for the
        'mirror' border of the 1-D convolution) and the planes are cropped on the device."""
        if level < 1:
            raise ValueError("recursive=True needs level >= 1")
        f64 = np.dtype(dtype) == np.float64 and _is_f64(arr)          # float64 engine (ref:319-320)
        arr = np.asarray(arr, dtype=np.float64 if f64 else np.float32)
        nd = arr.ndim
        ctx = default_context()
        fam = _taps_f64(scaling_function, nd) if f64 else _family_of(scaling_function, nd)
        acquire = acquire_plan64 if f64 else acquire_plan
        pad = (len(scaling_function.coefficients_1d) // 2) * 2 ** (level - 1)
        padded = np.pad(arr, pad, mode='symmetric')
        shape2 = {1: lambda a: (1, a.shape[0]), 2: lambda a: a.shape,
                  3: lambda a: (a.shape[0] * a.shape[1], a.shape[2])}[nd]
        big = acquire(ctx, *shape2(padded), fam, level)
        plan = acquire(ctx, *shape2(arr), fam, level)
        sym, conv_border = 1, (3 if nd == 1 else 1)   # symmetric / convolution() border, per sub-array
        try:
            big.upload(PLANE_INPUT, padded.reshape(shape2(padded)))
            if self.bilateral is None:
                big.set_border(conv_border)
                if nd == 3:
                    big.decompose3d(PLANE_INPUT, level, padded.shape[0])
                else:
                    big.decompose(PLANE_INPUT, level, 0)      # one kernel per scale
            else:
                sb = self._sigma_bilateral(level)
                cur = PLANE_INPUT
                for s in range(level):
                    nxt = level if s == level - 1 else PLANE_SCRATCH(s & 1)
                    f1 = float(sb[s]) ** 2
                    f2 = float(s + 1) if self.bilateral_scaling else 1.0
                    if nd == 3:
                        big.set_border(sym)
                        big.local_variance3d(cur, _TMP_PLANE, s, padded.shape[0], f1, f2)
                        big.bilateral3d_conv(cur, _TMP_PLANE, nxt, s, padded.shape[0])
                    else:
                        big.set_border(conv_border)           # ref:375: sdev_loc over convolution()
                        big.local_variance(cur, _TMP_PLANE, s, f1, f2)
                        big.set_border(sym)                   # ref:378: mode='symmetric'
                        big.bilateral_conv(cur, _TMP_PLANE, nxt, s,
                                           _lib.FLAG_TAPS_REVERSED if nd == 1 else 0)
                    big.binary("sub", cur, nxt, s)            # ref:402-403
                    cur = nxt
            for s in range(level + 1):                        # ref:405-406
                if nd == 3:
                    Z, Y, X = arr.shape
                    Yp = padded.shape[1]
                    for z in range(Z):
                        plan.copy_window_from(big, s, s, (z + pad) * Yp + pad, pad, z * Y, 0, Y, X)
                else:
                    plan.crop_from(big, s, s, 0 if nd == 1 else pad, pad)
        finally:
            big.set_border(0)
            release_plan(big)
        return Coefficients(plan, scaling_function, self.bilateral,
                            _shape=arr.shape if nd == 3 else None, _dtype=dtype)

    def _run(self, plan, level, src=PLANE_INPUT, flags=FLAG_FUSED):
        if self.bilateral is None:
            # (the first fused pass also histograms |w_0| for a get_noise() that may follow)
            plan.decompose(src, level, flags | (_lib.FLAG_MEDIAN_HIST if flags & FLAG_FUSED else 0))   # ref:432,442
        else:
            sb = self._sigma_bilateral(level)
            plan.decompose_bilateral(src, level, sb, self.bilateral_scaling, flags & ~FLAG_FUSED)

    def _sigma_bilateral(self, level):
        """Per-scale sigma_bilateral list (ref:421-424)."""
        sb = copy.copy(self.bilateral) if type(self.bilateral) is list \
            else [self.bilateral, ] * (level + 1)
        if len(sb) <= level:
            sb.extend([1, ] * (level - len(sb) + 1))
        return sb
