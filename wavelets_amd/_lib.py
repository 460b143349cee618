"""ctypes binding of libwatroo_hip.so (C ABI: include/watroo_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible the
import of the binding (or the first device call) raises.  numpy is the only dependency.
"""
import atexit
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WATROO_HIP_LIB", os.path.join(_HERE, "libwatroo_hip.so"))   # override: A/B builds

TRIANGLE, B3SPLINE = 0, 1
PLANE_INPUT, PLANE_OUT, PLANE_NONE = -1, -2, -1000
NUM_SCRATCH = 32
FLAG_FUSED, FLAG_NO_EXCHANGE, FLAG_SEPARATE_VARIANCE, FLAG_TAPS_REVERSED = 1, 2, 4, 8
FLAG_MEDIAN_HIST = 16     # wt_decompose_pass: the pass histograms |w_0| for the next wt_abs_median
PAD_POLY_SYMMETRIC, PAD_POLY_MIRROR = 5, 6     # wt_taps_conv_ex: the border rules of the recursive algorithm


def PLANE_SCRATCH(i):
    assert 0 <= i < NUM_SCRATCH
    return -3 - i


class WatrooHipError(RuntimeError):
    pass


_c = ctypes
_fp = _c.POINTER(_c.c_float)
_vp = _c.c_void_p
_i64 = _c.c_int64

# name -> (restype, argtypes); every symbol declared in include/watroo_hip.h
SIGNATURES = {
    "wt_abi_version": (_c.c_int, []),
    "wt_comm_version": (_c.c_int, [_c.POINTER(_c.c_int)]),
    "wt_unit_count": (_c.c_int, []),
    "wt_unit_name": (_c.c_char_p, [_c.c_int]),
    "wt_ctx_device_info": (_c.c_int, [_vp, _c.c_char_p, _c.c_int]),
    "wt_last_error": (_c.c_char_p, []),
    "wt_device_count": (_c.c_int, [_c.POINTER(_c.c_int)]),
    "wt_set_option": (_c.c_int, [_c.c_char_p, _c.c_int]),
    "wt_ctx_create": (_c.c_int, [_c.c_int, _c.POINTER(_vp)]),
    "wt_ctx_destroy": (_c.c_int, [_vp]),
    "wt_ctx_sync": (_c.c_int, [_vp]),
    "wt_device_memory": (_c.c_int, [_vp, _c.POINTER(_i64)]),
    "wt_timer_start": (_c.c_int, [_vp]),
    "wt_timer_stop": (_c.c_int, [_vp, _c.POINTER(_c.c_float)]),
    "wt_profile_enable": (_c.c_int, [_vp, _c.c_int]),
    "wt_profile_reset": (_c.c_int, [_vp]),
    "wt_profile_count": (_c.c_int, [_vp, _c.POINTER(_c.c_int)]),
    "wt_profile_entry": (_c.c_int, [_vp, _c.c_int, _c.c_char_p, _c.POINTER(_i64),
                                    _c.POINTER(_c.c_double)]),
    "wt_comm_unique_id": (_c.c_int, [_vp]),
    "wt_ctx_comm_init": (_c.c_int, [_vp, _c.c_int, _c.c_int, _vp]),
    "wt_ctx_comm_info": (_c.c_int, [_vp, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "wt_comm_selftest": (_c.c_int, [_vp, _i64, _c.POINTER(_c.c_int)]),
    "wt_plan_create": (_c.c_int, [_vp, _i64, _i64, _c.c_int, _c.c_int, _c.POINTER(_vp)]),
    "wt_plan_create_placed": (_c.c_int, [_vp, _i64, _i64, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_vp)]),
    "wt_plan_create_strip": (_c.c_int, [_vp, _i64, _i64, _c.c_int, _c.c_int, _i64, _i64, _i64,
                                        _c.c_int, _c.c_int, _c.POINTER(_vp)]),
    "wt_plan_destroy": (_c.c_int, [_vp]),
    "wt_plan_info": (_c.c_int, [_vp, _c.POINTER(_i64)]),
    "wt_plan_memory": (_c.c_int, [_vp, _c.POINTER(_i64)]),
    "wt_plan_trim": (_c.c_int, [_vp]),
    "wt_ctx_scatter_status": (_c.c_int, [_vp, _c.POINTER(_c.c_int), _c.c_char_p, _c.c_int]),
    "wt_schedule": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32), _c.c_int,
                               _c.POINTER(_c.c_int)]),
    "wt_plan_set_border": (_c.c_int, [_vp, _c.c_int]),
    "wt_plan_set_taps": (_c.c_int, [_vp, _fp, _c.c_int]),
    "wt_crop_plane": (_c.c_int, [_vp, _c.c_int, _vp, _c.c_int, _i64, _i64]),
    "wt_paste_plane": (_c.c_int, [_vp, _c.c_int, _vp, _c.c_int, _i64, _i64]),
    "wt_copy_window": (_c.c_int, [_vp, _c.c_int, _vp, _c.c_int] + [_i64] * 6),
    "wt_plane_ptr": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_vp)]),
    "wt_host_alloc": (_c.c_int, [_vp, _c.c_size_t, _c.POINTER(_vp)]),
    "wt_host_free": (_c.c_int, [_vp]),
    "wt_upload": (_c.c_int, [_vp, _c.c_int, _fp, _i64]),
    "wt_upload_int": (_c.c_int, [_vp, _c.c_int, _vp, _i64, _c.c_int]),
    "wt_download": (_c.c_int, [_vp, _c.c_int, _fp, _i64]),
    "wt_copy_plane": (_c.c_int, [_vp, _c.c_int, _c.c_int]),
    "wt_fill_plane": (_c.c_int, [_vp, _c.c_int, _c.c_float]),
    "wt_halo_exchange_local": (_c.c_int, [_vp, _vp, _c.c_int, _i64]),
    "wt_halo_exchange": (_c.c_int, [_vp, _c.c_int, _i64]),
    "wt_decompose": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt_decompose_pass": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt_decompose_sum": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt_decompose_pass_sum": (_c.c_int, [_vp] + [_c.c_int] * 8),
    "wt_decompose_sum_host": (_c.c_int, [_vp, _fp, _i64, _c.c_int, _c.c_int, _fp, _i64, _c.c_int]),
    "wt_denoise_sum_host": (_c.c_int, [_vp, _fp, _i64, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double),
                                       _c.POINTER(_c.c_double), _c.c_int, _c.c_int, _fp, _i64, _c.c_int]),
    "wt_plan_fused_ok": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_int)]),
    "wt_atrous_scale": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt_smooth": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt_local_variance": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float,
                                     _c.c_int, _c.c_int]),
    "wt_bilateral_conv": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt_decompose_bilateral": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.POINTER(_c.c_double),
                                          _c.c_int, _c.c_int]),
    "wt_plane_sum": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt_abs_median": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_float)]),
    "wt_significance": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int]),
    "wt_denoise": (_c.c_int, [_vp, _c.c_int, _c.c_double, _c.c_double, _c.c_int, _c.c_int]),
    "wt_denoise_sum": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                  _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.c_int,
                                  _c.c_int, _c.c_int]),
    "wt_wow_update": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int,
                                 _c.c_float, _c.c_int]),
    "wt_wow_scale": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int,
                                _c.c_float, _c.c_int, _c.c_int]),
    "wt_reduce": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_double)]),
    "wt_gamma_blend": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_float,
                                  _c.c_float]),
    "wt_smooth3d": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt_decompose3d": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt_local_variance3d": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    "wt_bilateral3d_conv": (_c.c_int, [_vp] + [_c.c_int] * 5),
    "wt_filter2d": (_c.c_int, [_vp, _c.c_int, _c.c_int, _fp, _c.c_int, _c.c_int, _c.c_int]),
    "wt_filter2d_ex": (_c.c_int, [_vp, _c.c_int, _c.c_int, _fp] + [_c.c_int] * 6),
    "wt_binary": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt_taps_conv": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32), _fp, _c.c_int,
                                _c.c_float, _c.c_int, _c.c_int, _c.c_int, _c.c_float]),
    "wt_taps_conv_ex": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32), _fp, _c.c_int,
                                   _c.c_float, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_int]),
    "wt_variance_from_moments": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_int]),
    "wt_mrs_update": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int,
                                 _c.c_int, _c.c_float]),
    "wt_anscombe": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_float,
                               _c.c_int]),
    # float64 engine (wt_plan64)
    "wt64_plan_create": (_c.c_int, [_vp, _i64, _i64, _c.c_int, _c.POINTER(_c.c_double), _c.c_int,
                                    _c.POINTER(_vp)]),
    "wt64_plan_destroy": (_c.c_int, [_vp]),
    "wt64_plan_set_border": (_c.c_int, [_vp, _c.c_int]),
    "wt64_upload": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_double), _i64]),
    "wt64_upload_int": (_c.c_int, [_vp, _c.c_int, _vp, _i64, _c.c_int]),
    "wt64_download": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_double), _i64]),
    "wt64_decompose": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_decompose_sum": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int)]),
    "wt64_smooth": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_local_variance": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_double,
                                       _c.c_double, _c.c_int, _c.c_int]),
    "wt64_bilateral_conv": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                       _c.c_int]),
    "wt_axis_filter": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_float), _c.c_int,
                                  _c.c_int, _c.c_int, _c.c_float, _c.c_int]),
    "wt64_axis_filter": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_double), _c.c_int,
                                    _c.c_int, _c.c_int, _c.c_double, _c.c_int]),
    "wt64_decompose_bilateral": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.POINTER(_c.c_double), _c.c_int]),
    "wt64_copy_window": (_c.c_int, [_vp, _c.c_int, _vp, _c.c_int, _i64, _i64, _i64, _i64, _i64,
                                    _i64]),
    "wt64_taps_conv": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32),
                                  _c.POINTER(_c.c_double), _c.c_int, _c.c_double, _c.c_int, _c.c_int, _c.c_int,
                                  _c.c_double]),
    "wt64_taps_conv_ex": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32),
                                     _c.POINTER(_c.c_double), _c.c_int, _c.c_double, _c.c_int, _c.c_int, _c.c_int,
                                     _c.c_double, _c.c_int]),
    "wt64_variance_from_moments": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_int]),
    "wt64_abs_median": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_double)]),
    "wt64_significance": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_int,
                                     _c.c_int, _c.c_int]),
    "wt64_plane_sum": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_plane_sum_early": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.POINTER(_c.c_int)]),
    "wt64_plane_sum_resume": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt_plane_sum_early": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.POINTER(_c.c_int)]),
    "wt_plane_sum_resume": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_binary": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_anscombe": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double,
                                 _c.c_int]),
    "wt64_wow_update": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int,
                                   _c.c_double, _c.c_int]),
    "wt64_wow_scale": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int, _c.c_double, _c.c_int]),
    "wt64_gamma_blend": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_double,
                                    _c.c_double, _c.c_double]),
    "wt64_fill_plane": (_c.c_int, [_vp, _c.c_int, _c.c_double]),
    "wt_fft_supported": (_c.c_int, [_i64, _i64, _c.POINTER(_c.c_int)]),
    "wt_fft_spectrum": (_c.c_int, [_vp, _c.c_int]),
    "wt_fft_apply": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_fft_spectrum": (_c.c_int, [_vp, _c.c_int]),
    "wt64_fft_apply": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_decompose_ex": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_plan_fused_ok": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_int)]),
    "wt64_decompose_pass": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_decompose_pass_sum": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int]),
    "wt64_denoise_sum": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double),
                                    _c.POINTER(_c.c_double), _c.c_int, _c.c_int, _c.c_int]),
    "wt64_reduce": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_double)]),
    "wt64_filter2d": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.POINTER(_c.c_double), _c.c_int,
                                 _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "wt64_mrs_update": (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int,
                                   _c.c_int, _c.c_double]),
}

_lib = None
_lock = threading.Lock()


def load():
    """Load libwatroo_hip.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise WatrooHipError(
                    f"{LIB_PATH} not found - the HIP engine is not built "
                    "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
            L = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
                fn.restype, fn.argtypes = res, args
            if L.wt_abi_version() != 8:
                raise WatrooHipError("libwatroo_hip.so ABI version mismatch")
            _lib = L
    return _lib


def comm_version():
    """ncclGetVersion of the RCCL library the engine loads (0: unknown); loads the library."""
    v = _c.c_int(0)
    check(load().wt_comm_version(_c.byref(v)))
    return v.value


def unit_names():
    """the translation units of the library's device code, as its warm-up threads know them (host logic)"""
    L = load()
    return [L.wt_unit_name(i).decode() for i in range(L.wt_unit_count())]


def check(rc):
    if rc != 0:
        raise WatrooHipError(load().wt_last_error().decode("utf-8", "replace"))


def fft_supported(H, W):
    """True when an H x W image can take the FFT path of the circular products (powers of two, 2 .. 8192)."""
    ok = _c.c_int(0)
    check(load().wt_fft_supported(H, W, _c.byref(ok)))
    return bool(ok.value)


_option_values = {"scatter": int(os.environ.get("WT_SCATTER", "4")),      # options restored after a temporary change
                  "scatter_strips": int(os.environ.get("WT_SCATTER_STRIPS", "0"))}


def set_option(name, value):
    check(load().wt_set_option(name.encode(), int(value)))
    if name in _option_values:
        _option_values[name] = int(value)


def device_count():
    n = _c.c_int(0)
    check(load().wt_device_count(_c.byref(n)))
    return n.value


class Context:
    """One GPU + one HIP stream (+ optional RCCL communicator).  wt_ctx."""

    def __init__(self, device=0):
        self._h = _vp()
        self.device = device
        self.rank, self.nranks = 0, 1
        L = load()
        if device_count() == 0:
            raise WatrooHipError("no HIP device visible: the a-trous engine needs an MI355X "
                                 "(there is no CPU fallback) ["
                                 + L.wt_last_error().decode("utf-8", "replace") + "]")
        check(L.wt_ctx_create(device, _c.byref(self._h)))
        _live.add(self)

    def close(self):
        if self._h:
            load().wt_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(load().wt_ctx_sync(self._h))

    def device_memory(self):
        """(free, total) bytes of the device (hipMemGetInfo)."""
        out = (_i64 * 2)()
        check(load().wt_device_memory(self._h, out))
        return int(out[0]), int(out[1])

    def scatter_status(self):
        """(disabled, reason): whether this context has fallen back from planes over scattered
        chunks to plain hipMalloc, and the HIP call that failed when it did."""
        d = _c.c_int(0)
        buf = _c.create_string_buffer(256)
        check(load().wt_ctx_scatter_status(self._h, _c.byref(d), buf, 256))
        return bool(d.value), buf.value.decode("utf-8", "replace")

    def timer_start(self):
        check(load().wt_timer_start(self._h))

    def timer_stop(self):
        ms = _c.c_float(0)
        check(load().wt_timer_stop(self._h, _c.byref(ms)))
        return ms.value

    def profile(self, on):
        check(load().wt_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(load().wt_profile_reset(self._h))

    def profile_entries(self):
        """{kernel name: (calls, total_ms)} measured with HIP events on the launch stream."""
        L = load()
        n = _c.c_int(0)
        check(L.wt_profile_count(self._h, _c.byref(n)))
        out = {}
        for i in range(n.value):
            name = _c.create_string_buffer(64)
            calls, ms = _i64(0), _c.c_double(0)
            check(L.wt_profile_entry(self._h, i, name, _c.byref(calls), _c.byref(ms)))
            out[name.value.decode()] = (calls.value, ms.value)
        return out

    # ---- RCCL
    @staticmethod
    def unique_id():
        buf = _c.create_string_buffer(128)
        check(load().wt_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, nranks, unique_id):
        assert len(unique_id) == 128
        buf = _c.create_string_buffer(unique_id, 128)
        check(load().wt_ctx_comm_init(self._h, rank, nranks, buf))
        self.rank, self.nranks = rank, nranks

    def comm_info(self):
        """(rank, size) as the RCCL communicator reports them; (0, 1) without one."""
        r, n = _c.c_int(0), _c.c_int(1)
        check(load().wt_ctx_comm_info(self._h, _c.byref(r), _c.byref(n)))
        return r.value, n.value

    def device_info(self):
        """{'device': ordinal, 'pci': bus id, 'cus': compute units, 'name': ...} of this context's GPU"""
        buf = _c.create_string_buffer(256)
        check(load().wt_ctx_device_info(self._h, buf, 256))
        text = buf.value.decode("utf-8", "replace")
        head, _, name = text.partition(" name=")
        info = dict(kv.split("=", 1) for kv in head.split())
        return {"device": int(info.get("device", -1)), "pci": info.get("pci", "?"), "cus": int(info.get("cus", 0)), "name": name}

    def comm_selftest(self, nfloats=1 << 20):
        ok = _c.c_int(0)
        check(load().wt_comm_selftest(self._h, nfloats, _c.byref(ok)))
        return bool(ok.value)


_default_ctx = {}
_live = weakref.WeakSet()        # Plans and Contexts still holding device resources


@atexit.register
def _shutdown():
    """Release device resources while the HIP runtime is still alive (interpreter teardown
    otherwise destroys planes after the runtime's own static destructors ran)."""
    _host_pool.close()
    objs = list(_live)
    for kind in ((Plan, Plan64), (Context,)):
        for o in objs:
            if isinstance(o, kind):
                try:
                    o.close()
                except Exception:           # (interpreter exit: report nothing, keep releasing)
                    pass
    _default_ctx.clear()
    _lane_ctx.clear()
    del _pool[:]


_tls = threading.local()          # .ctx: the context the API calls of THIS thread run on (use_context)
_lane_ctx = {}                    # device -> [Context]: the extra contexts of sequence.map_frames' worker lanes


class use_context:
    """`with use_context(ctx):` - the numpy-to-numpy calls of this thread (denoise, wow, AtrousTransform ...) run on
    `ctx` (its stream, its scratch, plans pooled under it) instead of the process-wide default context.  What the
    worker lanes of sequence.map_frames use: one context per lane, so that the upload of one frame, the passes of
    another and the download of a third overlap."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.prev = getattr(_tls, "ctx", None)
        _tls.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _tls.ctx = self.prev
        return False


def lane_contexts(n, device=None):
    """`n` contexts of their own (HIP stream, scratch, warm-up) on `device`, created once per process and reused."""
    if device is None:
        device = int(os.environ.get("WATROO_HIP_DEVICE", "0"))
    with _pool_lock:
        lanes = _lane_ctx.setdefault(device, [])
        while len(lanes) < n:
            lanes.append(Context(device))
        return lanes[:n]


def default_context(device=None):
    """Process-wide context (device from WATROO_HIP_DEVICE, default 0); inside `use_context(ctx)`: that context."""
    if device is None:
        ctx = getattr(_tls, "ctx", None)
        if ctx is not None and ctx._h:
            return ctx
        device = int(os.environ.get("WATROO_HIP_DEVICE", "0"))
    ctx = _default_ctx.get(device)
    if ctx is None:
        with _pool_lock:
            ctx = _default_ctx.get(device)
            if ctx is None:
                ctx = _default_ctx[device] = Context(device)
    return ctx


def schedule(family, level, fused=True):
    """Pass schedule [(first_scale, n_scales, halo_rows)] - host logic, needs no GPU."""
    tr = (_c.c_int32 * (3 * 32))()
    n = _c.c_int(0)
    check(load().wt_schedule(family, level, int(fused), tr, 32, _c.byref(n)))
    return [(tr[3 * i], tr[3 * i + 1], tr[3 * i + 2]) for i in range(n.value)]


class _HostPool:
    """Page-locked host blocks behind the ndarrays handed back to the caller.

    A fresh pageable `np.empty` target costs one page fault per 4 KiB during the device-to-host
    copy (8192^2 float32: 17-31 ms instead of 4.7 ms at PCIe rate).  Results >= 1 MiB are
    therefore allocated with wt_host_alloc and wrapped as ordinary writable ndarrays; when the
    last view of such an array dies its block returns to this pool and backs the next result
    of the same size.  WATROO_HIP_HOST_POOL_MB bounds the idle bytes kept (default 4096;
    0 disables the mechanism: plain np.empty)."""

    MIN_BYTES = 1 << 20

    def __init__(self):
        self.lock = threading.Lock()
        self.idle = {}            # nbytes -> [address]
        self.idle_bytes = 0
        self.limit = int(os.environ.get("WATROO_HIP_HOST_POOL_MB", "4096")) << 20
        self.closed = False

    def empty(self, ctx, shape, dtype=np.float32):
        shape = tuple(int(n) for n in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if self.limit == 0 or self.closed or nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype)
        nbytes = (nbytes + 65535) & ~65535
        addr = None
        with self.lock:
            lst = self.idle.get(nbytes)
            if lst:
                addr = lst.pop()
                self.idle_bytes -= nbytes
        if addr is None:
            ptr = _vp()
            if load().wt_host_alloc(ctx._h, nbytes, _c.byref(ptr)) != 0 or not ptr.value:
                return np.empty(shape, dtype)          # pinned memory exhausted: pageable
            addr = ptr.value
        block = (_c.c_char * nbytes).from_address(addr)
        weakref.finalize(block, self._give, addr, nbytes)
        n = int(np.prod(shape, dtype=np.int64))
        return np.frombuffer(block, dtype=dtype, count=n).reshape(shape)

    def _give(self, addr, nbytes):
        drop = []
        with self.lock:
            if self.closed:
                return                                  # interpreter exit: the OS reclaims it
            if nbytes > self.limit:
                drop.append(addr)
            else:
                while self.idle_bytes + nbytes > self.limit:
                    k = next(k for k, v in self.idle.items() if v)
                    drop.append(self.idle[k].pop())
                    self.idle_bytes -= k
                self.idle.setdefault(nbytes, []).append(addr)
                self.idle_bytes += nbytes
        for a in drop:
            load().wt_host_free(_vp(a))

    def close(self):
        with self.lock:
            self.closed = True
            blocks = [a for v in self.idle.values() for a in v]
            self.idle.clear()
            self.idle_bytes = 0
        for a in blocks:
            load().wt_host_free(_vp(a))


_host_pool = _HostPool()


def host_empty(shape, ctx=None, dtype=np.float32):
    """ndarray (float32 unless told otherwise) for a result coming back from the device (see _HostPool)."""
    return _host_pool.empty(ctx if ctx is not None else default_context(), shape, dtype)


# element types the device widens itself (wt_upload_int / wt64_upload_int): integers, bool (as uint8) and -
# in the other byte order only - float32 / float64 (FITS data is big-endian)
_ELEM_CODES = {"b1": 2, "i1": 1, "u1": 2, "i2": 3, "u2": 4, "i4": 5, "u4": 6, "i8": 7, "u8": 8, "f4": 9, "f8": 10}


def device_widens(dtype):
    """True for element types Plan.upload / Plan64.upload send over PCIe as they are."""
    dtype = np.dtype(dtype)
    if dtype.str[1:] not in _ELEM_CODES:
        return False
    return dtype.kind in "iub" or (dtype.kind == "f" and not dtype.isnative)


def _elem_source(host, shape):
    """(address, row pitch in bytes, type code) of a host image the device can widen, else None"""
    h = host
    if not device_widens(h.dtype) or h.ndim != 2 or h.shape != tuple(shape) or not h.size:
        return None
    if h.strides[1] != h.itemsize or h.strides[0] < h.shape[1] * h.itemsize:
        return None
    code = _ELEM_CODES[h.dtype.str[1:]]
    if not h.dtype.isnative and h.itemsize > 1:
        code |= 16                                       # WT_BYTESWAPPED
    return h.ctypes.data, h.strides[0], code


def _as_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("expected a 2-D image")
    return a


def _as_f32_rows(a):
    """float32 2-D array whose rows are contiguous (unit column stride, row stride a multiple of 4 bytes
    and >= the width): a row-strided view is handed to the library as it is (its row stride goes down
    as the host stride), anything else is copied"""
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("expected a 2-D image")
    if a.dtype == np.float32 and a.shape[1] > 0 and a.strides[1] == 4 and a.strides[0] % 4 == 0 \
            and a.strides[0] >= 4 * a.shape[1]:
        return a
    return np.ascontiguousarray(a, dtype=np.float32)


class Plan:
    """Device planes of one image (or one row strip of it).  wt_plan."""

    cold = False             # True: created by the pool (acquire_plan), planes on plain hipMalloc

    def __init__(self, ctx, H, W, family, max_level, row0=0, nrows=None, halo_rows=0,
                 rank=0, nranks=1, scatter=None):
        self._h = _vp()
        self.ctx = ctx
        nrows = H if nrows is None else nrows
        # `family` is TRIANGLE / B3SPLINE, or a tuple of 1-D taps for a user-defined scaling
        # function (wt_plan_set_taps); the tuple is kept as self.family (plan-pool key)
        taps = tuple(float(t) for t in family) if isinstance(family, (tuple, list)) else None
        if scatter is not None and nranks == 1 and row0 == 0 and nrows == H:
            # placement of THIS plan's planes (0: plain hipMalloc), whatever the process-wide "scatter" option says
            check(load().wt_plan_create_placed(ctx._h, H, W, B3SPLINE if taps else family, max_level, int(scatter),
                                               _c.byref(self._h)))
        else:
            check(load().wt_plan_create_strip(ctx._h, H, W, B3SPLINE if taps else family, max_level,
                                              row0, nrows, halo_rows, rank, nranks,
                                              _c.byref(self._h)))
        info = (_i64 * 8)()
        check(load().wt_plan_info(self._h, info))
        (self.H, self.W, self.pitch, self.row0, self.nrows, self.halo, self.max_level,
         self.family) = [int(v) for v in info]
        self.rank, self.nranks = rank, nranks
        _live.add(self)
        if taps:
            self.set_taps(taps)
            self.family = taps

    @property
    def custom(self):
        """True when the plan filters with user-defined taps (generic kernels, no fusion)."""
        return isinstance(self.family, tuple)

    def set_taps(self, taps):
        arr = (_c.c_float * max(len(taps), 1))(*taps)
        check(load().wt_plan_set_taps(self._h, arr, len(taps)))

    def close(self):
        if self._h:
            h, self._h = self._h, _vp()
            check(load().wt_plan_destroy(h))     # a failed release (leaked HBM) is not silent

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def memory(self):
        """(bytes held, bytes in scattered planes, idle chunk bytes, scatter disabled on this
        context) - wt_plan_memory."""
        out = (_i64 * 4)()
        check(load().wt_plan_memory(self._h, out))
        return int(out[0]), int(out[1]), int(out[2]), bool(out[3])

    def trim(self):
        check(load().wt_plan_trim(self._h))

    @property
    def shape(self):
        return (self.nrows, self.W)

    # ---- transfers
    def upload(self, plane, host):
        """plane <- host image as float32; integer and byte-swapped images cross PCIe as they are and
        are widened on the device (wt_upload_int) instead of by a host astype."""
        src = _elem_source(np.asarray(host), self.shape)
        if src is not None:
            check(load().wt_upload_int(self._h, plane, _vp(src[0]), src[1], src[2]))
            return
        host = _as_f32(host)
        if host.shape != self.shape:
            raise ValueError(f"image shape {host.shape} != plan strip shape {self.shape}")
        check(load().wt_upload(self._h, plane, host.ctypes.data_as(_fp), host.shape[1]))

    def download(self, plane, out=None):
        if out is None:
            out = host_empty(self.shape, self.ctx)
        assert out.dtype == np.float32 and out.shape == self.shape and out.strides[1] == 4
        check(load().wt_download(self._h, plane, out.ctypes.data_as(_fp), out.strides[0] // 4))
        return out

    def set_border(self, border):
        check(load().wt_plan_set_border(self._h, int(border)))

    def crop_from(self, src_plan, src_plane, dst_plane, y0, x0):
        check(load().wt_crop_plane(src_plan._h, src_plane, self._h, dst_plane, y0, x0))

    def copy_window_from(self, src_plan, src_plane, dst_plane, sy, sx, dy, dx, rows, cols):
        """self[dst_plane][dy:dy+rows, dx:dx+cols] = src_plan[src_plane][sy:sy+rows, sx:sx+cols]"""
        check(load().wt_copy_window(src_plan._h, src_plane, self._h, dst_plane, sy, sx, dy, dx,
                                    rows, cols))

    def paste_into(self, dst_plan, src_plane, dst_plane, y0, x0):
        check(load().wt_paste_plane(self._h, src_plane, dst_plan._h, dst_plane, y0, x0))

    def copy(self, src, dst):
        check(load().wt_copy_plane(self._h, src, dst))

    def fill(self, plane, value):
        check(load().wt_fill_plane(self._h, plane, value))

    def plane_ptr(self, plane):
        p = _vp()
        check(load().wt_plane_ptr(self._h, plane, _c.byref(p)))
        return p.value

    def halo_exchange(self, plane, rows):
        check(load().wt_halo_exchange(self._h, plane, rows))

    @staticmethod
    def halo_exchange_local(upper, lower, plane, rows):
        check(load().wt_halo_exchange_local(upper._h, lower._h, plane, rows))

    # ---- hot path
    def decompose(self, src, level, flags=FLAG_FUSED):
        check(load().wt_decompose(self._h, src, level, flags))

    def decompose_sum(self, src, level, dst=PLANE_OUT, flags=FLAG_FUSED):
        """decompose + plane sum in the same passes (bit-identical to the two calls)."""
        check(load().wt_decompose_sum(self._h, src, level, dst, flags))

    def decompose_sum_host(self, host, level, dst=PLANE_OUT, out=None, block_rows=0):
        """upload + decompose_sum + download(dst) with the PCIe legs pipelined behind the passes
        (wt_decompose_sum_host): returns the reconstruction as an ndarray; the device state is
        that of the three calls."""
        host = _as_f32_rows(host)                 # (a row-strided view goes down as it is: hipMemcpy2D)
        if host.shape != self.shape:
            raise ValueError(f"image shape {host.shape} != plan strip shape {self.shape}")
        if out is None:
            out = host_empty(self.shape, self.ctx)
        assert out.dtype == np.float32 and out.shape == self.shape and out.strides[1] == 4
        check(load().wt_decompose_sum_host(self._h, host.ctypes.data_as(_fp), host.strides[0] // 4, level, dst,
                                           out.ctypes.data_as(_fp), out.strides[0] // 4, block_rows))
        return out

    def denoise_sum_host(self, host, level, k_passes, taus, wgts, soft, dst=PLANE_OUT, out=None, block_rows=0):
        """upload + first k_passes passes + denoise_sum over their planes + remaining passes +
        download(dst), pipelined over blocks of rows (wt_denoise_sum_host): the denoised image."""
        host = _as_f32_rows(host)
        if host.shape != self.shape:
            raise ValueError(f"image shape {host.shape} != plan strip shape {self.shape}")
        if out is None:
            out = host_empty(self.shape, self.ctx)
        assert out.dtype == np.float32 and out.shape == self.shape and out.strides[1] == 4
        n = len(taus)
        t = (_c.c_double * n)(*[float(v) for v in taus])
        w = (_c.c_double * n)(*[float(v) for v in wgts])
        check(load().wt_denoise_sum_host(self._h, host.ctypes.data_as(_fp), host.strides[0] // 4, level, k_passes, n, t, w,
                                         int(soft), dst, out.ctypes.data_as(_fp), out.strides[0] // 4, block_rows))
        return out

    def decompose_pass_sum(self, cur, nxt, s0, ns, flags, sum_plane, first, last):
        check(load().wt_decompose_pass_sum(self._h, cur, nxt, s0, ns, flags, sum_plane,
                                           int(first), int(last)))

    def fused_ok(self, level):
        """True when decompose_sum(level) runs as accumulate passes (wt_plan_fused_ok)."""
        ok = _c.c_int(0)
        check(load().wt_plan_fused_ok(self._h, level, _c.byref(ok)))
        return bool(ok.value)

    def decompose_pass(self, cur, nxt, s0, ns, flags=FLAG_FUSED):
        check(load().wt_decompose_pass(self._h, cur, nxt, s0, ns, flags))

    def atrous_scale(self, src, dst_c, dst_w, s, flags=0):
        check(load().wt_atrous_scale(self._h, src, dst_c, dst_w, s, flags))

    def smooth(self, src, dst, s, square_input=False, flags=0):
        check(load().wt_smooth(self._h, src, dst, s, int(square_input), flags))

    def local_variance(self, src, dst, s, f1=1.0, f2=1.0, take_sqrt=False, flags=0):
        check(load().wt_local_variance(self._h, src, dst, s, f1, f2, int(take_sqrt), flags))

    def bilateral_conv(self, src, var, dst, s, flags=0):
        check(load().wt_bilateral_conv(self._h, src, var, dst, s, flags))

    def decompose_bilateral(self, src, level, sigma_b, bilateral_scaling=False, flags=0):
        arr = (_c.c_double * max(level, 1))(*[float(v) for v in sigma_b[:level]])
        check(load().wt_decompose_bilateral(self._h, src, level, arr, int(bilateral_scaling),
                                            flags))

    def plane_sum(self, first, count, dst=PLANE_OUT):
        check(load().wt_plane_sum(self._h, first, count, dst))

    def plane_sum_early(self, count, dst=PLANE_OUT):
        """planes [0, count) summed into dst on the side stream, if the plan is in the overlapped state of a
        bilateral transform (True), else nothing (False)"""
        done = _c.c_int(0)
        check(load().wt_plane_sum_early(self._h, count, dst, _c.byref(done)))
        return bool(done.value)

    def plane_sum_resume(self, first, count, dst=PLANE_OUT):
        check(load().wt_plane_sum_resume(self._h, first, count, dst))

    def abs_median(self, plane):
        m = _c.c_float(0)
        check(load().wt_abs_median(self._h, plane, _c.byref(m)))
        return np.float32(m.value)

    def significance(self, plane, dst, tau, soft=True, noise_plane=PLANE_NONE):
        check(load().wt_significance(self._h, plane, dst, float(tau), int(soft), noise_plane))

    def denoise(self, plane, tau, wgt=1.0, soft=True, noise_plane=PLANE_NONE):
        check(load().wt_denoise(self._h, plane, float(tau), float(wgt), int(soft), noise_plane))

    def denoise_sum(self, count, taus, wgts, soft=True, noise_plane=PLANE_NONE, write_back=False,
                    dst=PLANE_OUT, first=0):
        n = len(taus)
        ta = (_c.c_double * max(n, 1))(*[float(t) for t in taus])
        wa = (_c.c_double * max(n, 1))(*[float(w) for w in wgts])
        check(load().wt_denoise_sum(self._h, first, count, dst, n, ta, wa, int(soft), noise_plane,
                                    int(write_back)))

    def wow_update(self, plane, power_plane, tau, soft, noise_plane, factor, gamma_plane):
        check(load().wt_wow_update(self._h, plane, power_plane, float(tau), int(soft),
                                   noise_plane, float(factor), gamma_plane))

    def wow_scale(self, plane, s, tau, soft, noise_plane, factor, gamma_plane, flags=0):
        check(load().wt_wow_scale(self._h, plane, s, float(tau), int(soft), noise_plane,
                                  float(factor), gamma_plane, flags))

    def reduce(self, plane):
        """(sum, sumsq, min, max) over the GLOBAL image, fp64."""
        out = (_c.c_double * 4)()
        check(load().wt_reduce(self._h, plane, out))
        return tuple(out)

    def gamma_blend(self, recon, gamma_plane, gmin, gmax, inv_gamma, h):
        check(load().wt_gamma_blend(self._h, recon, gamma_plane, gmin, gmax, inv_gamma, h))

    def smooth3d(self, src, dst, s, depth):
        check(load().wt_smooth3d(self._h, src, dst, s, depth))

    def local_variance3d(self, src, dst, s, depth, f1=1.0, f2=1.0):
        check(load().wt_local_variance3d(self._h, src, dst, s, depth, f1, f2))

    def bilateral3d_conv(self, src, var, dst, s, depth):
        check(load().wt_bilateral3d_conv(self._h, src, var, dst, s, depth))

    def decompose3d(self, src, level, depth):
        check(load().wt_decompose3d(self._h, src, level, depth))

    def filter2d(self, src, dst, kernel, flags=0, anchor=None, periodic=False):
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        if k.ndim != 2:
            raise ValueError("filter2d kernel must be 2-D")
        ay, ax = (k.shape[0] // 2, k.shape[1] // 2) if anchor is None else anchor
        check(load().wt_filter2d_ex(self._h, src, dst, k.ctypes.data_as(_fp), k.shape[0],
                                    k.shape[1], ay, ax, 3 if periodic else 0, flags))

    def binary(self, op, a, b, dst):
        check(load().wt_binary(self._h, {"sub": 0, "add": 1, "mul": 2, "div": 3,
                                         "add_div": 4}[op], a, b, dst))

    def taps_conv(self, src, var, dst, offsets, weights, center_weight=None, depth=0, pad_mode=0,
                  fill_value=0.0, dilation=1):
        """generic tap-list operator (wt_taps_conv_ex): offsets (n, 3) int32 = (dz, dy, dx); dilation:
        the stride of the polyphase pad modes (5 / 6)"""
        offs = np.ascontiguousarray(offsets, dtype=np.int32).reshape(-1, 3)
        wts = np.ascontiguousarray(weights, dtype=np.float32).ravel()
        assert len(offs) == len(wts)
        check(load().wt_taps_conv_ex(self._h, src, var, dst, offs.ctypes.data_as(_c.POINTER(_c.c_int32)),
                                     wts.ctypes.data_as(_fp), len(wts),
                                     0.0 if center_weight is None else float(center_weight),
                                     int(center_weight is not None), depth, pad_mode, float(fill_value),
                                     int(dilation)))

    def axis_filter(self, src, dst, axis, offsets, weights, depth=0, pad_mode=0, fill_value=0.0, dilation=1):
        """K-tap filter along one axis (2 = x, 1 = y, 0 = z) on the tiled kernels (wt_axis_filter)"""
        offs = np.ascontiguousarray(offsets, dtype=np.int32).ravel()
        wts = np.ascontiguousarray(weights, dtype=np.float32).ravel()
        assert len(offs) == len(wts)
        check(load().wt_axis_filter(self._h, src, dst, axis, offs.ctypes.data_as(_c.POINTER(_c.c_int32)), wts.ctypes.data_as(_fp),
                                    len(wts), depth, pad_mode, float(fill_value), int(dilation)))

    def variance_from_moments(self, mean, meansq, dst, f1=1.0, f2=1.0, take_sqrt=False):
        """sdev_loc's last step (ref wavelets.py:27-32) from conv(I) and conv(I^2)"""
        check(load().wt_variance_from_moments(self._h, mean, meansq, dst, f1, f2, int(take_sqrt)))

    def mrs_update(self, plane, mrs_plane, tau, soft, noise_plane, persistent, inv_pow):
        check(load().wt_mrs_update(self._h, plane, mrs_plane, float(tau), int(soft), noise_plane,
                                   int(persistent), float(inv_pow)))

    def fft_spectrum(self, src):
        """kernel spectrum of the plan <- FFT2 of plane src (wt_fft_spectrum)"""
        check(load().wt_fft_spectrum(self._h, src))

    def fft_apply(self, src, dst, conj=False):
        """dst = irfft2(rfft2(src) * K) or, conj, * conj(K) (wt_fft_apply)"""
        check(load().wt_fft_apply(self._h, src, dst, int(conj)))

    def anscombe(self, src, dst, alpha=1.0, g=0.0, sigma=0.0, inverse=False):
        check(load().wt_anscombe(self._h, src, dst, alpha, g, sigma, int(inverse)))


# ------------------------------------------------------------------------------------------
# plan pool: hipMalloc/hipFree of a dozen multi-hundred-MiB planes costs milliseconds, far more
# than the transform itself; plans of recently used geometries are kept for reuse.
# ------------------------------------------------------------------------------------------
_POOL_MAX_BYTES = int(os.environ.get("WATROO_HIP_POOL_BYTES", str(16 << 30)))
_pool = []            # [(key, plan)] most recently released last


def _plan_bytes(plan):
    """Device bytes a pooled plan holds: the library's own count for float32 plans (planes, the
    bounce plane of host transfers, idle chunks), an estimate for float64 plans."""
    if isinstance(plan, Plan) and plan._h:
        return plan.memory()[0]
    return (plan.nrows + 2 * plan.halo) * plan.pitch * 8 * (plan.max_level + 1 + 2 + 4)


_pool_lock = threading.RLock()      # plan pool and default contexts are shared by host threads


_dp = _c.POINTER(_c.c_double)


class Plan64:
    """Double-precision planes of one image: wt_plan64, the float64 engine (the reference computes
    float64 / promoted inputs in float64, ref wavelets.py:297,319-320).  Same plane ids and, for the
    operators it has, the same method names as ``Plan``; taps are always given (the built-in
    families pass theirs).  Standard decomposition without bilateral filtering, Coefficients
    operators, convolution, sdev_loc, Anscombe."""

    dtype = np.float64
    custom = True            # generic kernels with run-time taps: no fused passes
    rank, nranks, row0, halo = 0, 1, 0, 0

    def __init__(self, ctx, H, W, taps, max_level):
        self._h = _vp()
        self.ctx = ctx
        self.family = tuple(float(t) for t in taps)
        arr = (_c.c_double * len(self.family))(*self.family)
        check(load().wt64_plan_create(ctx._h, H, W, max_level, arr, len(self.family),
                                      _c.byref(self._h)))
        self.H, self.W, self.nrows, self.max_level = H, W, H, max_level
        self.pitch = (W + 1) // 2 * 2
        _live.add(self)

    def close(self):
        if self._h:
            load().wt64_plan_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shape(self):
        return (self.H, self.W)

    _INT_CODES = _ELEM_CODES          # (emptied by tools/bench_int_input.py to time the host promotion)
    device_widens = staticmethod(device_widens)

    def upload(self, plane, host):
        """plane <- host image as float64.  Integer and big-endian images (what the reference recasts to
        float64 first, ref wavelets.py:297, 319-320) cross PCIe as they are and are widened on the device."""
        src = _elem_source(np.asarray(host), self.shape) if self._INT_CODES else None
        if src is not None:
            check(load().wt64_upload_int(self._h, plane, _vp(src[0]), src[1], src[2]))
            return
        host = np.ascontiguousarray(host, dtype=np.float64)
        if host.shape != self.shape:
            raise ValueError(f"image shape {host.shape} != plan shape {self.shape}")
        check(load().wt64_upload(self._h, plane, host.ctypes.data_as(_dp), host.shape[1]))

    def download(self, plane, out=None):
        if out is None:
            out = host_empty(self.shape, self.ctx, np.float64)
        assert out.dtype == np.float64 and out.shape == self.shape and out.strides[1] == 8
        check(load().wt64_download(self._h, plane, out.ctypes.data_as(_dp), out.strides[0] // 8))
        return out

    def set_border(self, border):
        check(load().wt64_plan_set_border(self._h, int(border)))

    _BUILTIN = {(0.25, 0.5, 0.25): TRIANGLE, (0.0625, 0.25, 0.375, 0.25, 0.0625): B3SPLINE}

    @property
    def fused_family(self):
        """TRIANGLE / B3SPLINE when the taps are a built-in family's (fused passes), else None"""
        return self._BUILTIN.get(self.family)

    def fused_ok(self, level):
        """True when the whole schedule of `level` scales runs as fused passes (wt64_plan_fused_ok)."""
        ok = _c.c_int(0)
        check(load().wt64_plan_fused_ok(self._h, level, _c.byref(ok)))
        return bool(ok.value)

    def decompose(self, src, level, flags=0):
        check(load().wt64_decompose_ex(self._h, src, level, 0, flags & FLAG_MEDIAN_HIST))

    def decompose_pass(self, cur, nxt, s0, ns, flags=FLAG_FUSED):
        check(load().wt64_decompose_pass(self._h, cur, nxt, s0, ns, flags & FLAG_MEDIAN_HIST))

    def decompose_pass_sum(self, cur, nxt, s0, ns, flags, sum_plane, first, last):
        check(load().wt64_decompose_pass_sum(self._h, cur, nxt, s0, ns, sum_plane, int(first), int(last)))

    def decompose_sum(self, src, level, dst=PLANE_OUT, flags=0):
        """Transform + np.sum(planes, axis=0) -> dst; True when the sum rode in the fused passes."""
        fused = _c.c_int(0)
        check(load().wt64_decompose_sum(self._h, src, level, dst, _c.byref(fused)))
        return bool(fused.value)

    def decompose3d(self, src, level, depth):
        check(load().wt64_decompose(self._h, src, level, depth))

    def smooth(self, src, dst, s, square_input=False, flags=0):
        check(load().wt64_smooth(self._h, src, dst, s, int(square_input), 0))

    def smooth3d(self, src, dst, s, depth):
        check(load().wt64_smooth(self._h, src, dst, s, 0, depth))

    def local_variance(self, src, dst, s, f1=1.0, f2=1.0, take_sqrt=False, flags=0):
        check(load().wt64_local_variance(self._h, src, dst, s, f1, f2, int(take_sqrt), 0))

    def local_variance3d(self, src, dst, s, depth, f1=1.0, f2=1.0):
        check(load().wt64_local_variance(self._h, src, dst, s, f1, f2, 0, depth))

    def bilateral_conv(self, src, var, dst, s, flags=0):
        check(load().wt64_bilateral_conv(self._h, src, var, dst, s, 0,
                                         int(bool(flags & FLAG_TAPS_REVERSED))))

    def bilateral3d_conv(self, src, var, dst, s, depth):
        check(load().wt64_bilateral_conv(self._h, src, var, dst, s, depth, 0))

    def decompose_bilateral(self, src, level, sigma_b, bilateral_scaling=False, flags=0):
        arr = (_c.c_double * max(level, 1))(*[float(v) for v in sigma_b[:level]])
        check(load().wt64_decompose_bilateral(self._h, src, level, arr, int(bilateral_scaling)))

    def copy_window_from(self, src_plan, src_plane, dst_plane, sy, sx, dy, dx, rows, cols):
        check(load().wt64_copy_window(src_plan._h, src_plane, self._h, dst_plane, sy, sx, dy, dx,
                                      rows, cols))

    def crop_from(self, src_plan, src_plane, dst_plane, y0, x0):
        self.copy_window_from(src_plan, src_plane, dst_plane, y0, x0, 0, 0, self.H, self.W)

    def abs_median(self, plane):
        m = _c.c_double(0)
        check(load().wt64_abs_median(self._h, plane, _c.byref(m)))
        return np.float64(m.value)

    def significance(self, src, dst, tau, soft=True, noise_plane=PLANE_NONE):
        check(load().wt64_significance(self._h, src, dst, tau, 1.0, int(soft), noise_plane, 0))

    def denoise(self, plane, tau, weight=1.0, soft=True, noise_plane=PLANE_NONE):
        check(load().wt64_significance(self._h, plane, plane, tau, weight, int(soft), noise_plane, 1))

    def wow_update(self, plane, power_plane, tau, soft, noise_plane, factor, gamma_plane):
        check(load().wt64_wow_update(self._h, plane, power_plane, float(tau), int(soft),
                                     noise_plane, float(factor), gamma_plane))

    def wow_scale(self, plane, s, tau, soft, noise_plane, factor, gamma_plane, flags=0):
        """local power conv_s(c^2) + the wow update of one scale, in place (wt64_wow_scale)"""
        check(load().wt64_wow_scale(self._h, plane, s, float(tau), int(soft), noise_plane, float(factor), gamma_plane))

    def gamma_blend(self, recon, gamma_plane, gmin, gmax, inv_gamma, h):
        check(load().wt64_gamma_blend(self._h, recon, gamma_plane, gmin, gmax, inv_gamma, h))

    def fill(self, plane, value):
        check(load().wt64_fill_plane(self._h, plane, value))

    def reduce(self, plane):
        out = (_c.c_double * 4)()
        check(load().wt64_reduce(self._h, plane, out))
        return tuple(out)

    def denoise_sum(self, n, taus, wgts, soft, noise_plane=PLANE_NONE, write_back=True, dst=PLANE_OUT):
        """Coefficients.denoise over the first len(taus) planes fused with the plane sum of planes
        0..n-1 -> dst (wt64_denoise_sum); tau <= 0: no threshold on that plane (weight only)."""
        k = len(taus)
        t = (_c.c_double * max(k, 1))(*[float(v) for v in taus])
        w = (_c.c_double * max(k, 1))(*[float(v) for v in wgts])
        check(load().wt64_denoise_sum(self._h, 0, n, dst, k, t, w, int(soft), noise_plane, int(write_back)))

    def plane_sum(self, first, count, dst=PLANE_OUT):
        check(load().wt64_plane_sum(self._h, first, count, dst))

    def plane_sum_early(self, count, dst=PLANE_OUT):
        done = _c.c_int(0)
        check(load().wt64_plane_sum_early(self._h, count, dst, _c.byref(done)))
        return bool(done.value)

    def plane_sum_resume(self, first, count, dst=PLANE_OUT):
        check(load().wt64_plane_sum_resume(self._h, first, count, dst))

    def binary(self, op, a, b, dst):
        code = {"add": 0, "sub": 1, "mul": 2, "div": 3, "add_div": 4}[op]
        check(load().wt64_binary(self._h, code, a, b, dst))

    def taps_conv(self, src, var, dst, offsets, weights, center_weight=None, depth=0, pad_mode=0,
                  fill_value=0.0, dilation=1):
        offs = np.ascontiguousarray(offsets, dtype=np.int32).reshape(-1, 3)
        wts = np.ascontiguousarray(weights, dtype=np.float64).ravel()
        assert len(offs) == len(wts)
        check(load().wt64_taps_conv_ex(self._h, src, var, dst, offs.ctypes.data_as(_c.POINTER(_c.c_int32)),
                                       wts.ctypes.data_as(_dp), len(wts),
                                       0.0 if center_weight is None else float(center_weight),
                                       int(center_weight is not None), depth, pad_mode, float(fill_value),
                                       int(dilation)))

    def axis_filter(self, src, dst, axis, offsets, weights, depth=0, pad_mode=0, fill_value=0.0, dilation=1):
        offs = np.ascontiguousarray(offsets, dtype=np.int32).ravel()
        wts = np.ascontiguousarray(weights, dtype=np.float64).ravel()
        assert len(offs) == len(wts)
        check(load().wt64_axis_filter(self._h, src, dst, axis, offs.ctypes.data_as(_c.POINTER(_c.c_int32)), wts.ctypes.data_as(_dp),
                                      len(wts), depth, pad_mode, float(fill_value), int(dilation)))

    def variance_from_moments(self, mean, meansq, dst, f1=1.0, f2=1.0, take_sqrt=False):
        check(load().wt64_variance_from_moments(self._h, mean, meansq, dst, f1, f2, int(take_sqrt)))

    def copy(self, src, dst):
        self.copy_window_from(self, src, dst, 0, 0, 0, 0, self.H, self.W)

    def fft_spectrum(self, src):
        check(load().wt64_fft_spectrum(self._h, src))

    def fft_apply(self, src, dst, conj=False):
        check(load().wt64_fft_apply(self._h, src, dst, int(conj)))

    def filter2d(self, src, dst, kernel, flags=0, anchor=None, periodic=False):
        k = np.ascontiguousarray(kernel, dtype=np.float64)
        if k.ndim != 2:
            raise ValueError("filter2d kernel must be 2-D")
        ay, ax = (k.shape[0] // 2, k.shape[1] // 2) if anchor is None else anchor
        check(load().wt64_filter2d(self._h, src, dst, k.ctypes.data_as(_dp), k.shape[0], k.shape[1],
                                   ay, ax, 3 if periodic else 0))

    def mrs_update(self, plane, mrs_plane, tau, soft, noise_plane, persistent, inv_pow):
        check(load().wt64_mrs_update(self._h, plane, mrs_plane, float(tau), int(soft), noise_plane,
                                     int(persistent), float(inv_pow)))

    def anscombe(self, src, dst, alpha=1, g=0, sigma=0, inverse=False):
        check(load().wt64_anscombe(self._h, src, dst, alpha, g, sigma, int(inverse)))


def acquire_plan64(ctx, H, W, taps, max_level):
    """Pooled float64 plan (same pool and eviction rule as the float32 plans)."""
    taps = tuple(float(t) for t in taps)
    key = (id(ctx), H, W, ("f64",) + taps, max_level)
    with _pool_lock:
        for i in range(len(_pool) - 1, -1, -1):
            if _pool[i][0] == key:
                plan = _pool.pop(i)[1]
                plan.set_border(0)
                return plan
    return Plan64(ctx, H, W, taps, max_level)


_COLD_FIRST_USE = not os.environ.get("WATROO_HIP_NO_COLD_PLANS")


def acquire_plan(ctx, H, W, family, max_level):
    """A whole-image plan for (H, W, family, max_level): pooled if available, else new.

    Plans made HERE - for the numpy-to-numpy calls of the API - keep their planes on plain hipMalloc (round 5).
    Mapping planes over scattered physical chunks makes the fused passes ~20 % faster, 0.1 ms
    per transform at 8192^2, and costs ~16 ms when a plan is created and first used: worth it for device-resident
    loops over a `_lib.Plan` (bench.py's metric: input already in HBM), not for calls that move the image over PCIe
    both ways (10 ms at 8192^2) - and a one-shot `denoise(img)` would pay it in full (section 3.8: first call of a
    process 48 -> 38 ms).  WATROO_HIP_NO_COLD_PLANS=1 restores scattered planes for pooled plans too."""
    key = (id(ctx), H, W, family, max_level)
    with _pool_lock:
        for i in range(len(_pool) - 1, -1, -1):
            if _pool[i][0] == key:
                plan = _pool.pop(i)[1]
                plan.set_border(0)
                return plan
    if not _COLD_FIRST_USE:
        return Plan(ctx, H, W, family, max_level)
    plan = Plan(ctx, H, W, family, max_level, scatter=0)      # (per plan: no process-wide option is touched)
    plan.cold = True
    return plan


def release_plan(plan):
    """Hand a plan back for reuse (its planes keep stale data; callers re-upload)."""
    if plan is None or not plan._h or plan.nranks != 1:
        return
    evicted = []
    fam = ("f64",) + plan.family if isinstance(plan, Plan64) else plan.family
    if isinstance(plan, Plan):
        try:
            plan.trim()           # idle physical chunks (up to three planes' worth) go back now
        except WatrooHipError:
            # a plan whose chunks cannot be released is not pooled (it would be handed out again with
            # its state unknown): closed here, and the error goes to the caller
            try:
                plan.close()
            except WatrooHipError:
                pass
            raise
    with _pool_lock:
        _pool.append(((id(plan.ctx), plan.H, plan.W, fam, plan.max_level), plan))
        total = sum(_plan_bytes(p) for _, p in _pool)
        while _pool and (total > _POOL_MAX_BYTES or len(_pool) > 8):
            _, old = _pool.pop(0)
            total -= _plan_bytes(old)
            evicted.append(old)
    first_error = None
    for old in evicted:           # every evicted plan is closed, whatever the ones before it did
        try:
            old.close()
        except WatrooHipError as e:
            first_error = first_error or e
    if first_error is not None:
        raise first_error
