"""ctypes binding of libttnet.so (include/ttnet.h).

There is no fallback: if the library is missing or a call fails, this raises.  The only
torch objects that reach the C ABI are raw ``data_ptr()`` addresses and the current HIP
stream handle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libttnet.so")

TTNET_F32, TTNET_I64, TTNET_U8, TTNET_U16, TTNET_U64 = 0, 1, 2, 3, 4
VARIANTS = {"small": 0, "xsmall": 1, "full": 2, "valexnet": 3}


class NetDesc(C.Structure):
    _fields_ = [("variant", C.c_int32), ("nfilter", C.c_int32), ("tfilter", C.c_int32),
                ("layers", C.c_int32), ("image_h", C.c_int32), ("image_w", C.c_int32),
                ("max_batch", C.c_int32), ("reserved", C.c_int32)]


class TTNetError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libttnet status {status}: {message}")
        self.status = status


# every symbol include/ttnet.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS: List[Tuple[str, object, list]] = [
    ("ttnet_plan_create", C.c_int, [C.POINTER(NetDesc), C.c_int, C.POINTER(_P)]),
    ("ttnet_plan_set_tensor", C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int]),
    ("ttnet_plan_finalize", C.c_int, [_P, _P]),
    ("ttnet_forward", C.c_int, [_P, _P, C.c_int64, _P, _P]),
    ("ttnet_plan_set_lanes", C.c_int, [_P, C.c_int]),
    ("ttnet_forward_lane", C.c_int, [_P, C.c_int, _P, C.c_int64, _P, _P]),
    ("ttnet_plan_set_input_norm", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("ttnet_forward_u8", C.c_int, [_P, C.c_int, _P, C.c_int64, _P, _P]),
    ("ttnet_resize_center_crop_u8", C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    ("ttnet_forward_from_stem_bits", C.c_int, [_P, _P, C.c_int64, _P, _P]),
    ("ttnet_read_stage", C.c_int, [_P, C.c_char_p, C.c_int64, _P, C.c_size_t, C.c_int, _P]),
    ("ttnet_plan_get_table", C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    ("ttnet_plan_set_table", C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    ("ttnet_plan_query", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    ("ttnet_plan_set_profiling", C.c_int, [_P, C.c_int]),
    ("ttnet_plan_last_timings", C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
    ("ttnet_plan_destroy", None, [_P]),
    ("ttnet_comm_unique_id", C.c_int, [_P]),
    ("ttnet_comm_create", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("ttnet_allgather_logits", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    ("ttnet_comm_destroy", None, [_P]),
    ("ttnet_last_error", C.c_char_p, []),
    ("ttnet_version", C.c_char_p, []),
]

_lib = None


def load() -> C.CDLL:
    """Load libttnet.so (once).  Raises if it has not been built: there is no other path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m scale_imagenet_amd.build` "
            "(hipcc, gfx950).  scale_imagenet_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> int:
    if status < 0:
        raise TTNetError(status, load().ttnet_last_error().decode("utf-8", "replace"))
    return status
