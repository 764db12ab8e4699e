"""ctypes wrapper over oracle/fsq_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        _lib.fsq_constants.argtypes = [i32p, ctypes.c_int, f32p, f32p, f32p, f32p, i32p]
        _lib.fsq_forward.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, i32p, f32p, i32p, f32p]
        _lib.fsq_backward.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int, i32p, f32p]
        _lib.fsq_indices_to_codes.argtypes = [i32p, ctypes.c_int64, ctypes.c_int, i32p, f32p]
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def constants(levels):
    lv = np.ascontiguousarray(levels, dtype=np.int32)
    d = lv.size
    out = [np.empty(d, dtype=np.float32) for _ in range(4)] + [np.empty(d, dtype=np.int32)]
    lib().fsq_constants(_i(lv), d, _f(out[0]), _f(out[1]), _f(out[2]), _f(out[3]), _i(out[4]))
    return dict(zip(("half_l", "offset", "shift", "half_width", "basis"), out))


def forward(z, levels):
    """z (N,d) fp32 -> codes (N,d) fp32, indices (N,) int32, bounded (N,d) fp32 (pre-rounding values)."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    lv = np.ascontiguousarray(levels, dtype=np.int32)
    codes = np.empty_like(z)
    bounded = np.empty_like(z)
    idx = np.empty(z.shape[0], dtype=np.int32)
    lib().fsq_forward(_f(z), z.shape[0], z.shape[1], _i(lv), _f(codes), _i(idx), _f(bounded))
    return codes, idx, bounded


def backward(z, dcodes, levels):
    z = np.ascontiguousarray(z, dtype=np.float32)
    dcodes = np.ascontiguousarray(dcodes, dtype=np.float32)
    lv = np.ascontiguousarray(levels, dtype=np.int32)
    dz = np.empty_like(z)
    lib().fsq_backward(_f(z), _f(dcodes), z.shape[0], z.shape[1], _i(lv), _f(dz))
    return dz


def indices_to_codes(indices, levels):
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    lv = np.ascontiguousarray(levels, dtype=np.int32)
    codes = np.empty((idx.size, lv.size), dtype=np.float32)
    lib().fsq_indices_to_codes(_i(idx), idx.size, lv.size, _i(lv), _f(codes))
    return codes
