"""ctypes binding of libspp_hip.so (C ABI declared in include/superpoint.h).

Built in-tree by ``python -m onepose_amd.build_ext`` (hipcc, gfx950).  No fallback: if the shared
object is missing, ``load()`` raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_void_p

from ._native import NativeError
from .build_ext import SPP_LIB_PATH as LIB_PATH

NUM_LAYERS = 12
LAYER_NAMES = ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b",
               "convPa", "convPb", "convDa", "convDb")
KERNEL_IDS = {"conv1a": 0, "conv1b": 1, "pool": 2, "conv2": 3, "conv3a": 4, "conv3b": 5, "conv4": 6, "heads": 7,
              "convPb": 8, "convDb": 9, "score_map": 10, "nms": 11, "rowcount": 12, "rowscan": 13, "compact": 14,
              "select": 15, "cellnorm": 16, "sample": 17, "rank": 18, "scatter": 19}


class RawWeights(ctypes.Structure):
    """struct spp_raw_weights (device pointers, forward order)."""
    _fields_ = [("weight", c_void_p * NUM_LAYERS), ("bias", c_void_p * NUM_LAYERS)]


FLAG_PREC_FP16X4 = 0x800
PRECISIONS = {"fp32": 0, "fp16x4": FLAG_PREC_FP16X4}   # arithmetic of the GEMM convolutions, a `flags` bit per call

_FWD = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
        c_void_p, c_void_p, c_size_t, c_void_p, c_int]

# name -> (restype, argtypes); every symbol include/superpoint.h declares
SYMBOLS = {
    "spp_version": (c_int, []),
    "spp_last_error": (c_char_p, []),
    "spp_packed_weights_bytes": (c_size_t, []),
    "spp_pack_weights": (c_int, [POINTER(RawWeights), c_void_p, c_void_p]),
    "spp_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "spp_dense": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
    "spp_detect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "spp_forward": (c_int, _FWD),
    "spp_forward_profiled": (c_int, _FWD + [c_int, c_int, c_void_p, c_void_p]),
}

_lib = None


def load():
    """dlopen the HIP library and bind every entry point.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the SuperPoint HIP extension has not been built "
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
        msg = load().spp_last_error()
        raise NativeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
