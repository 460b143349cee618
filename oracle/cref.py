"""ctypes loader for the C restatement (oracle/atrous_ref.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by the
product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WT_ORACLE_LIB: another build of the same source (tools/oracle_sanitize.sh: -fsanitize=address,undefined)
_SO = os.environ.get("WT_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")
FAMILY = {"triangle": 0, "b3spline": 1}
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "atrous_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        L.orc_smooth.argtypes = [fp, fp, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int]
        L.orc_decompose.argtypes = [fp, fp, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                    ctypes.c_int]
        L.orc_plane_sum.argtypes = [fp, ctypes.c_int, ctypes.c_long, fp]
        L.orc_abs_median.argtypes = [fp, ctypes.c_long, fp]
        L.orc_denoise.argtypes = [fp, ctypes.c_long, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_int]
        L.orc_variance.argtypes = [fp, fp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_int]
        L.orc_bilateral.argtypes = [fp, fp, fp, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                    ctypes.c_int]
        L.orc_decompose_bilateral.argtypes = [fp, fp, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                              ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                              ctypes.c_int]
        L.orc_clip_sqrt.argtypes = [fp, ctypes.c_long]
        L.orc_scale_div.argtypes = [fp, fp, ctypes.c_float, ctypes.c_long]
        L.orc_scale.argtypes = [fp, ctypes.c_float, ctypes.c_long]
        L.orc_axpy1.argtypes = [fp, fp, ctypes.c_long]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().orc_num_threads()


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def set_threads(n):
    lib().orc_set_threads(int(n))


def smooth(img, family, s, square_input=False):
    img = _c32(img)
    out = np.empty_like(img)
    rc = lib().orc_smooth(_p(img), _p(out), img.shape[0], img.shape[1], FAMILY[family], s,
                          int(square_input))
    assert rc == 0
    return out


def decompose(img, level, family="b3spline"):
    img = _c32(img)
    planes = np.empty((level + 1,) + img.shape, np.float32)
    rc = lib().orc_decompose(_p(img), _p(planes), img.shape[0], img.shape[1], FAMILY[family],
                             level)
    assert rc == 0
    return planes


def plane_sum(planes):
    planes = _c32(planes)
    out = np.empty(planes.shape[1:], np.float32)
    lib().orc_plane_sum(_p(planes), planes.shape[0], out.size, _p(out))
    return out


def abs_median(x):
    x = _c32(x)
    out = np.zeros(1, np.float32)
    lib().orc_abs_median(_p(x), x.size, _p(out))
    return out[0]


def denoise_plane(plane, tau, wgt=1.0, soft=True):
    """in place on a C-contiguous float32 plane"""
    assert plane.dtype == np.float32 and plane.flags.c_contiguous
    lib().orc_denoise(_p(plane), plane.size, float(tau), float(wgt), int(soft))
    return plane


# ---------------------------------------------------------------------------------------------
# bilateral transform and utils.wow (BASELINE config 5) - same restatement as atrous_numpy.py,
# heavy loops in C/OpenMP so that 8192^2 x 11 scales finishes in tens of seconds
# ---------------------------------------------------------------------------------------------
def sigma_e(family, bilateral=None):
    """2-D sigma_e tables (watroo/wavelets.py:245-254, 274-283) - data, shared with the numpy oracle"""
    from . import atrous_numpy as O
    return O.sigma_e(family, bilateral, 2)


def variance(img, family, s, f1=1.0, f2=None):
    """sdev_loc(img, variance=True) * f1 (* f2) - watroo/wavelets.py:24-32, :434-436"""
    img = _c32(img)
    out = np.empty_like(img)
    rc = lib().orc_variance(_p(img), _p(out), img.shape[0], img.shape[1], FAMILY[family], s,
                            float(f1), float(f2 if f2 is not None else 1.0), int(f2 is not None))
    assert rc == 0
    return out


def bilateral(img, var, family, s):
    """atrous_convolution(img, kernel, bilateral_variance=var, s) - watroo/wavelets.py:74-105"""
    img, var = _c32(img), _c32(np.broadcast_to(var, np.shape(img)))
    out = np.empty_like(img)
    rc = lib().orc_bilateral(_p(img), _p(var), _p(out), img.shape[0], img.shape[1],
                             FAMILY[family], s)
    assert rc == 0
    return out


def decompose_bilateral(img, level, family, bilateral, bilateral_scaling=False):
    """atrous_standard with bilateral (scalar or list) - watroo/wavelets.py:408-444"""
    from . import atrous_numpy as O
    img = _c32(img)
    sb = O._sigma_bilateral_list(bilateral, level)
    arr = (ctypes.c_double * max(level, 1))(*[float(v) for v in sb[:level]])
    planes = np.empty((level + 1,) + img.shape, np.float32)
    rc = lib().orc_decompose_bilateral(_p(img), _p(planes), img.shape[0], img.shape[1],
                                       FAMILY[family], level, arr, int(bool(bilateral_scaling)))
    assert rc == 0
    return planes


def wow(data, family="b3spline", n_scales=None, weights=[], whitening=True,
        denoise_coefficients=[], noise=None, bilateral=None, bilateral_scaling=False,
        soft_threshold=True, preserve_variance=False, gamma=3.2, gamma_min=None,
        gamma_max=None, h=0):
    """utils.wow (watroo/utils.py:105-219) for a 2-D float32 ndarray and a scalar noise: the
    plumbing of atrous_numpy.wow statement for statement, every full-plane loop in C.  Global
    scalars (np.std, np.mean, min, max) stay numpy calls on the planes: their pairwise float32
    summation is part of what is being restated.  Returns (image, planes)."""
    import copy
    from . import atrous_numpy as O
    data = _c32(data)
    assert data.ndim == 2 and (noise is None or np.ndim(noise) == 0)
    L_ = lib()
    n_scales = O.wow_n_scales(data.shape, family, n_scales, h, denoise_coefficients, bilateral)
    sigma_bilateral = None if bilateral is None else O._sigma_bilateral_list(bilateral, n_scales)
    if bilateral is None:
        planes = decompose(data, n_scales, family)
    else:
        planes = decompose_bilateral(data, n_scales, family, sigma_bilateral, bilateral_scaling)
    sig_e = O.sigma_e(family, sigma_bilateral, 2)
    npix = data.size
    if h > 0:
        gamma_scaled = np.zeros_like(data)
    rw = copy.copy(weights)
    if len(rw) <= n_scales:
        rw.extend([1, ] * (n_scales - len(rw) + 1))
    sdc = copy.copy(denoise_coefficients)
    if len(sdc) < n_scales:
        sdc.extend([0, ] * (n_scales - len(sdc)))
    if len(sdc) == n_scales:
        sdc.extend([1, ])
    lp = np.empty_like(data)
    for s, (c, w, d) in enumerate(zip(planes, rw, sdc)):
        if preserve_variance:
            power_norm = np.std(c) if s == n_scales else np.sqrt(np.mean(c ** 2))
        else:
            power_norm = 1
        if s == n_scales:
            if whitening and h < 1:
                local_power = np.std(c)
                if local_power <= 0:
                    local_power = 1e-15
            else:
                local_power = 1
            if h > 0:
                L_.orc_axpy1(_p(gamma_scaled), _p(c), npix)
            c *= w * power_norm / local_power                      # scalars: as utils.py:203
            continue
        have_lp = whitening and h < 1
        if have_lp:                                                # utils.py:193-196 (power = c**2
            L_.orc_smooth(_p(c), _p(lp), c.shape[0], c.shape[1], FAMILY[family], s, 1)   # pre-threshold)
            L_.orc_clip_sqrt(_p(lp), npix)
        if d != 0:                                                 # utils.py:199, wavelets.py:129-143
            if noise is None:
                noise = np.median(np.abs(planes[0])) / 0.6745 / sig_e[0]
            if noise != 0:
                L_.orc_denoise(_p(c), npix, float(d * noise * sig_e[s]), 1.0, int(bool(soft_threshold)))
        if h > 0:
            L_.orc_axpy1(_p(gamma_scaled), _p(c), npix)
        if have_lp:
            L_.orc_scale_div(_p(c), _p(lp), np.float32(w * power_norm), npix)
        else:
            c *= w * power_norm / 1
    recon = plane_sum(planes)
    if h > 0:
        if gamma_min is None:
            gamma_min = gamma_scaled.min()
        if gamma_max is None:
            gamma_max = gamma_scaled.max()
        gamma_scaled -= gamma_min
        gamma_scaled /= gamma_max - gamma_min
        gamma_scaled[gamma_scaled < 0] = 0
        gamma_scaled[gamma_scaled > 1] = 1
        gamma_scaled **= 1 / gamma
        recon = (1 - h) * recon + h * gamma_scaled
    return recon, planes
