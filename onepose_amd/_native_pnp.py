"""ctypes binding of libpnp_hip.so (C ABI declared in include/pnp.h).  No fallback: a missing library raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_size_t, c_uint64, c_void_p

from ._native import NativeError
from .build_ext import PNP_LIB_PATH as LIB_PATH

# name -> (restype, argtypes); every symbol include/pnp.h declares
SYMBOLS = {
    "pnp_version": (c_int, []),
    "pnp_last_error": (c_char_p, []),
    "pnp_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pnp_ransac_epnp": (c_int, [c_void_p, c_void_p, POINTER(c_double), c_double, c_int, c_double, c_int, c_uint64, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pnp_ransac_epnp_matches": (c_int, [c_void_p, c_void_p, c_void_p, c_int, POINTER(c_double), c_double, c_double, c_int, c_uint64,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pnp_epnp": (c_int, [c_void_p, c_void_p, POINTER(c_double), c_double, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the PnP HIP extension has not been built "
            "(run `python -m onepose_amd.build_ext`; needs hipcc).  There is no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pnp_last_error()
        raise NativeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
