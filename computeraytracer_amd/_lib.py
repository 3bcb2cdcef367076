"""ctypes binding of libcrt.so (the C ABI declared in include/crt.h).

There is no fallback of any kind: if the shared library is missing or a HIP
device is not available, loading / crt_create raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRT_LIB") or os.path.join(_HERE, "libcrt.so")   # CRT_LIB: tuning builds only

NCOUNTERS = 8
ACCEL_NONE, ACCEL_BVH2, ACCEL_LBVH = 0, 1, 2
CNT = dict(rays=0, nodes=1, prims=2, paths=3, bounces=4, shadow=5, hits=6, walked=7)

# name -> (restype, argtypes); kept in one place so tests can check that every
# symbol include/crt.h declares is exported.
_P = C.c_void_p
SIGNATURES = {
    "crt_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "crt_destroy": (None, [_P]),
    "crt_last_error": (C.c_char_p, [_P]),
    "crt_abi_version": (C.c_int, []),
    "crt_upload_scene": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, _P, _P]),
    "crt_set_tile": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "crt_set_row_bands": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "crt_build_accel": (C.c_int, [_P, C.c_int]),
    "crt_reset": (C.c_int, [_P]),
    "crt_trace": (C.c_int, [_P, C.c_uint32]),
    "crt_sync": (C.c_int, [_P]),
    "crt_sample_count": (C.c_int, [_P, _P]),
    "crt_tile": (C.c_int, [_P, _P]),
    "crt_read_accum": (C.c_int, [_P, _P]),
    "crt_read_rgba8": (C.c_int, [_P, _P]),
    "crt_write_accum": (C.c_int, [_P, _P, C.c_uint32]),
    "crt_read_latest_rgba8": (C.c_int, [_P, _P, _P]),
    "crt_latest_sample": (C.c_int, [_P, _P]),
    "crt_pin_host": (C.c_int, [_P, C.c_size_t]),
    "crt_unpin_host": (C.c_int, [_P]),
    "crt_read_sample_rgba8": (C.c_int, [_P, C.c_uint32, _P]),
    "crt_device_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "crt_bind_output": (C.c_int, [_P, _P, _P]),
    "crt_set_stream": (C.c_int, [_P, _P]),
    "crt_enable_counters": (C.c_int, [_P, C.c_int]),
    "crt_counters": (C.c_int, [_P, _P]),
    "crt_reset_counters": (C.c_int, [_P]),
    "crt_last_trace_ms": (C.c_int, [_P, _P, _P]),
    "crt_last_kernel_ms": (C.c_int, [_P, _P, _P]),
    "crt_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "crt_accel_stats": (C.c_int, [_P, _P]),
    "crt_get_device": (C.c_int, [_P, _P]),
    "crt_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "crt_image_size": (C.c_int, [_P, _P]),
    "crt_comm_unique_id": (C.c_int, [_P, C.c_int]),
    "crt_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "crt_comm_partition": (C.c_int, [_P, C.c_uint32]),
    "crt_gather": (C.c_int, [_P, C.c_int]),
    "crt_read_frame_rgba8": (C.c_int, [_P, _P]),
    "crt_read_frame_accum": (C.c_int, [_P, _P]),
    "crt_frame_device_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "crt_comm_info": (C.c_int, [_P, _P]),
    "crt_comm_destroy": (C.c_int, [_P]),
    "crt_layout_rows": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P]),
    "crt_debug_intersect": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "crt_debug_probes": (C.c_int, [_P, _P]),
    "crt_debug_math": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_size_t]),
}

_lib = None


class CrtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libcrt error {code}: {message}")
        self.code = code


def load():
    """dlopen libcrt.so and bind every entry point. Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C computeraytracer_amd/csrc)")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
