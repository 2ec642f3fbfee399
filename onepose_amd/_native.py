"""ctypes binding of libgatsspg_hip.so (C ABI declared in include/gatsspg.h).

The library is built in-tree by ``python -m onepose_amd.build_ext`` (hipcc, gfx950).  There is no
fallback: if the shared object is missing, ``load()`` raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_void_p

from .build_ext import LIB_PATH

NUM_GATS, NUM_ATTN = 4, 8
FLAG_INCLUDE_SELF, FLAG_ADDITIONAL, FLAG_WITH_LINEAR_TRANSFORM = 1, 2, 4
FLAG_PREC_BF16X3 = 0x100
FLAG_PREC_BF16X6 = 0x200
FLAG_PREC_FP16X3 = 0x400
FLAG_PREC_FP16X4 = 0x800
PRECISIONS = {"fp32": 0, "bf16x3": FLAG_PREC_BF16X3, "bf16x6": FLAG_PREC_BF16X6, "fp16x3": FLAG_PREC_FP16X3, "fp16x4": FLAG_PREC_FP16X4}   # GEMM arithmetic of the attention layers, selected per call
KERNEL_IDS = {"load_state": 0, "gats": 1, "qkv_kv": 2, "kv_final": 3, "mlp0": 5, "stat_final": 6,
              "mlp3": 7, "final_proj_norm": 8, "score_exp": 9, "conf_finalize": 10, "match_tail": 11, "gats_wlt": 12,
              "softmax_stats": 13}
LAYER_SELF, LAYER_CROSS = 0, 1


class RawWeights(ctypes.Structure):
    """struct gatsspg_raw_weights (device pointers)."""
    _fields_ = [
        ("gats_W", c_void_p * NUM_GATS), ("gats_a", c_void_p * NUM_GATS),
        ("proj_w", (c_void_p * 3) * NUM_ATTN), ("proj_b", (c_void_p * 3) * NUM_ATTN),
        ("merge_w", c_void_p * NUM_ATTN), ("merge_b", c_void_p * NUM_ATTN),
        ("mlp0_w", c_void_p * NUM_ATTN), ("mlp0_b", c_void_p * NUM_ATTN),
        ("mlp3_w", c_void_p * NUM_ATTN), ("mlp3_b", c_void_p * NUM_ATTN),
        ("final_w", c_void_p), ("final_b", c_void_p),
    ]


class KencWeights(ctypes.Structure):
    """struct gatsspg_kenc_weights (device pointers)."""
    _fields_ = [("w", c_void_p * 4), ("b", c_void_p * 4), ("inp_dim", c_int)]


# name -> (restype, argtypes); every symbol include/gatsspg.h declares
SYMBOLS = {
    "gatsspg_version": (c_int, []),
    "gatsspg_last_error": (c_char_p, []),
    "gatsspg_packed_weights_bytes": (c_size_t, []),
    "gatsspg_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gatsspg_pack_weights": (c_int, [POINTER(RawWeights), c_void_p, c_void_p]),
    "gatsspg_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gatsspg_forward_profiled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_size_t, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gatsspg_db_cache_bytes": (c_size_t, [c_int, c_int]),
    "gatsspg_prepare_database": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                         c_void_p, c_size_t, c_void_p]),
    "gatsspg_forward_cached": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int,
                                       c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "gatsspg_load_state": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "gatsspg_store_state": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "gatsspg_gats_layer": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                   c_void_p]),
    "gatsspg_attn_layer": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "gatsspg_final_proj_norm": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "gatsspg_score_dual_softmax_match": (c_int, [c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gatsspg_kenc_scratch_bytes": (c_size_t, [c_int, c_int]),
    "gatsspg_keypoint_encoder": (c_int, [POINTER(KencWeights), c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                         c_size_t, c_void_p]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def load():
    """dlopen the HIP library and bind every entry point.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the GATsSPG HIP extension has not been built "
            "(run `python -m onepose_amd.build_ext`; needs hipcc).  There is no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gatsspg_last_error()
        raise NativeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
