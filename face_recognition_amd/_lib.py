"""ctypes binding of libface_recognition_amd.so (the C ABI declared in include/face_recognition_amd.h).

There is no fallback: if the shared object is missing or a call fails, a
``RuntimeError`` carrying the library's own message is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRA_LIB_PATH: another build of the library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("FRA_LIB_PATH") or os.path.join(_HERE, "libface_recognition_amd.so")

FRA_F32, FRA_BF16, FRA_F16 = 0, 1, 2
DTYPE_BY_NAME = {"f32": FRA_F32, "float32": FRA_F32, "bf16": FRA_BF16, "bfloat16": FRA_BF16,
                 "f16": FRA_F16, "float16": FRA_F16, "fp16": FRA_F16}


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


# name -> (restype, argtypes); every symbol include/face_recognition_amd.h declares
_P, _I, _L, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
SIGNATURES = {
    "fra_last_error": (C.c_char_p, []),
    "fra_version": (_I, []),
    "fra_device_fault": (_I, [_I, C.POINTER(C.c_uint)]),
    "fra_dbg_set_spin_limit": (_I, [_I]),
    "fra_set_small_batch_mode": (_I, [_I]),
    "fra_get_small_batch_mode": (_I, []),
    "fra_dbg_alloc_calls": (C.c_long, []),
    "mfn_create": (_I, [C.POINTER(_P), _I, _I, _I]),
    "mfn_load_weights": (_I, [_P, C.POINTER(TensorDesc), _I]),
    "mfn_workspace_bytes": (_I, [_P, _I, C.POINTER(C.c_size_t)]),
    "mfn_set_batch_chunk": (_I, [_P, _I]),
    "mfn_set_fused": (_I, [_P, _I]),
    "mfn_set_chain": (_I, [_P, _I]),
    "mfn_launch_plan": (_I, [_P, _I, C.POINTER(_I), _I]),
    "mfn_forward": (_I, [_P, _P, _I, _P, _P, C.c_size_t, _P]),
    "mfn_forward_ex": (_I, [_P, _P, _I, _I, _I, _P, _P, C.c_size_t, _P]),
    "mfn_workspace_bytes_ex": (_I, [_P, _I, _I, C.POINTER(C.c_size_t)]),
    "mfn_forward_timed": (_I, [_P, _P, _I, _P, _P, C.c_size_t, _P, C.POINTER(C.c_float), _I]),
    "mfn_op_count": (_I, [_P]),
    "mfn_op_variant": (_I, [_P, _I, _I]),
    "mfn_op_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_I), C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)]),
    "mfn_debug_prefix": (_I, [_P, _P, _I, _I, _P, C.POINTER(_L), _P, C.c_size_t, _P]),
    "mfn_destroy": (_I, [_P]),
    "mfn_tta_combine": (_I, [_P, _P, _I, _I, _P, _P]),
    "mfn_op_stem": (_I, [_I, _P, _P, _P, _P, _P, _I, _P]),
    "mfn_op_dw3x3": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mfn_op_pw": (_I, [_I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "mfn_op_block": (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mfn_rblock_package_bytes": (_L, [_I, _I]),
    "mfn_pack_rblock": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfn_op_rblock": (_I, [_I, _I, _I, _I, _P, _P, _I, _P, _I, _P]),
    "mfn_op_wblock": (_I, [_I, _I, _I, _I, _P, _P, _I, _P, _I, _P]),
    "mfn_op_gdconv7": (_I, [_I, _P, _P, _P, _P, _I, _I, _P]),
    "mfn_op_tail": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mfn_op_linear_l2norm": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fb_create": (_I, [C.POINTER(_P), _I, _P, _I, _L, _I, _I, _L]),
    "fb_fill": (_I, [_P, _L, _P, _I, _L]),
    "fb_fill_async": (_I, [_P, _L, _P, _L, _P]),
    "fb_rows": (_L, [_P]),
    "fb_reserve": (_I, [_P, _I]),
    "fb_export_packed": (_I, [_P, _L, _L, _P, _I, _P, _I]),
    "fb_import_packed": (_I, [_P, _L, _L, _P, _I, _P, _I]),
    "fb_search_l2_topk": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "fb_search_l2_topk_noalloc": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "fb_merge_topk": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "fb_threshold_score": (_I, [_P, _P, _P, _I, _D, _P]),
    "fb_match_nearest": (_I, [_P, _P, _I, _D, _P, _P, _P, _P]),
    "fb_destroy": (_I, [_P]),
    "ev_fold_scores": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "fa_align_warp": (_I, [_P, _I, _I, _P, _I, _P, _I, _I, _P]),
}

_lib: Optional[C.CDLL] = None


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and may not have loaded them yet when this
    library is.  Loaded on its own, this library would then pull in the system copies (/opt/rocm/lib), the process would
    hold TWO HIP runtimes, and whichever initialises second reports "no ROCm-capable device".  Loading torch's copies
    first makes the dynamic loader resolve this library's libamdhip64.so.N to them (same SONAME)."""
    try:
        import torch
    except ImportError:
        return
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load() -> C.CDLL:
    """Load the shared object once and attach signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m face_recognition_amd.build` "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().fra_last_error()
        raise RuntimeError(f"{what} failed (status {rc}): {msg.decode() if msg else 'no message'}")


def current_stream_ptr(device) -> int:
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)
