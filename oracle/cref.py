"""ctypes loader for the C restatement (oracle/atrous_ref.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by the
product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
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
