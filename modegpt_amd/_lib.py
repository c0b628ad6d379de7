"""ctypes binding of libmodegpt_hip.so (include/modegpt_hip.h).

There is no CPU fallback: if the shared library is missing the import of any
compute entry point raises, and every non-zero status becomes a RuntimeError
carrying mdg_last_error().
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmodegpt_hip.so")

MDG_OK = 0
MDG_ERR_BAD_ARG, MDG_ERR_HIP, MDG_ERR_NOT_PD, MDG_ERR_NO_CONVERGE, MDG_ERR_NO_DEVICE = -1, -2, -3, -4, -5
MDG_BF16, MDG_F16, MDG_F32, MDG_F64 = 0, 1, 2, 3
MDG_QK_ROPE_GROUPED, MDG_QK_ROPE_MHA, MDG_QK_OPT = 0, 1, 2
MDG_I8_NO_EXACT, MDG_I8_EXACT_ALWAYS = 1, 2
MDG_GEMM_LOWER_ONLY, MDG_GEMM_A_LOWER_TRI, MDG_GEMM_B_LOWER_TRI, MDG_GEMM_A_UPPER_TRI = 1, 2, 4, 8

_i64, _i32, _f64, _ptr, _sz = C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_size_t


class CovProblem(C.Structure):
    """mdg_cov_problem"""
    _fields_ = [("x", _ptr), ("n_tokens", _i64), ("n_feat", _i64), ("batch", _i64), ("ld", _i64), ("sigma", _ptr),
                ("ld_sigma", _i64), ("sigma_batch_stride", _i64)]


ABI_VERSION = 9   # include/modegpt_hip.h MDG_ABI_VERSION this binding table was written against

# name -> (restype, argtypes); must list every symbol the header declares (tests/test_abi.py checks it)
SIGNATURES = {
    "mdg_abi_version": (_i32, []),
    "mdg_last_error": (C.c_char_p, []),
    "mdg_device_info": (_i32, [_i32, C.c_char_p, _i32, C.POINTER(_i32), C.POINTER(_i64)]),
    "mdg_shutdown": (_i32, []),
    "mdg_deferred_status_begin": (_i32, [_ptr, _ptr]),
    "mdg_deferred_status_end": (_i32, []),
    "mdg_deferred_status_decode": (_i32, [C.POINTER(_i32)]),
    "mdg_cov_accum_ws_bytes": (_sz, [_i64, _i64, _i64]),
    "mdg_cov_accum": (_i32, [_ptr, _i32, _i64, _i64, _i64, _i64, _i32, _ptr, _i64, _i64, _ptr, _sz, _ptr]),
    "mdg_cov_accum_i8_ws_bytes": (_sz, [_i64, _i64]),
    "mdg_cov_accum_i8": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _sz, _f64, _i32, C.POINTER(_i32), _ptr, _ptr, _ptr, _ptr]),
    "mdg_cov_accum_i8_stats": (_i32, [_ptr, _i64, _i64, C.POINTER(C.c_ulonglong), _ptr]),
    "mdg_cov_accum_i8_route": (_i32, [_i32, C.POINTER(CovProblem), _i32, _ptr, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                                       C.POINTER(_f64), C.POINTER(_i32), _ptr]),
    "mdg_cov_accum_i8_multi_ws_bytes": (_sz, [_i32, C.POINTER(CovProblem)]),
    "mdg_cov_accum_i8_multi": (_i32, [_i32, C.POINTER(CovProblem), _ptr, _sz, _f64, _i32, C.POINTER(_i32), _ptr, _ptr, _ptr, _ptr]),
    "mdg_cov_accum_multi_ws_bytes": (_sz, [_i32, C.POINTER(CovProblem), _i32]),
    "mdg_cov_accum_multi": (_i32, [_i32, C.POINTER(CovProblem), _i32, _ptr, _sz, _ptr]),
    "mdg_cov_finalize": (_i32, [_ptr, _i64, _i64, _i64, _i64, _f64, _ptr]),
    "mdg_bi_ws_bytes": (_sz, [_i64]),
    "mdg_bi_accum": (_i32, [_ptr, _ptr, _i32, _i64, _i64, _i64, _ptr, _ptr, _sz, _ptr]),
    "mdg_gemm_f64": (_i32, [_i64, _i64, _i64, _f64, _ptr, _i32, _i64, _i64, _ptr, _ptr, _i32, _i64, _i64, _f64,
                             _ptr, _i32, _i64, _i64, _i64, _i64, _i64, _i32, _ptr]),
    "mdg_potrf_inv_diag_elems": (_sz, [_i64]),
    "mdg_potrf_lower": (_i32, [_ptr, _i64, _i64, _ptr, _ptr]),
    "mdg_potrs_lower_ws_bytes": (_sz, [_i64, _i64]),
    "mdg_potrs_lower": (_i32, [_ptr, _i64, _i64, _ptr, _ptr, _i64, _i64, _ptr, _sz, _ptr]),
    "mdg_chol_inverse_diag_ws_bytes": (_sz, [_i64]),
    "mdg_chol_inverse_diag": (_i32, [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _sz, _ptr]),
    "mdg_syevj_batched": (_i32, [_ptr, _i64, _i64, _ptr, _ptr, _ptr]),
    "mdg_ridge_scores_ws_bytes": (_sz, [_i64]),
    "mdg_ridge_scores": (_i32, [_ptr, _i64, _i64, _f64, _ptr, _ptr, _ptr, _sz, _ptr]),
    "mdg_select_smallest_sorted": (_i32, [_ptr, _i64, _i64, _ptr, _ptr]),
    "mdg_select_margin": (_i32, [_ptr, _ptr, _ptr, _i64, _i64, _f64, _ptr, _ptr]),
    "mdg_gather_rows_16": (_i32, [_ptr, _i64, _ptr, _i64, _i64, _ptr, _i64, _ptr]),
    "mdg_nystrom_down_ws_bytes": (_sz, [_i64, _i64, _i64]),
    "mdg_nystrom_down": (_i32, [_ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _i64, _i32, _f64, _ptr, _i64, _ptr, _ptr, _sz,
                                 _ptr]),
    "mdg_nystrom_down_overlapped": (_i32, [_ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _i64, _i32, _f64, _ptr, _i64, _ptr, _ptr, _sz,
                                            _ptr, _ptr, _ptr, _ptr]),
    "mdg_qk_select": (_i32, [_ptr, _ptr, _i32, _i32, _i32, _f64, _f64, _i32, _i32, _ptr, _ptr, _ptr, _ptr]),
    "mdg_vo_compress_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "mdg_vo_compress": (_i32, [_ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _i32, _i32, _i32, _i32, _i32, _f64, _ptr, _i64,
                                _ptr, _i64, _ptr, _ptr, _ptr, _sz, _ptr]),
    "mdg_sqrt_psd_small_ws_bytes": (_sz, [_i64, _i64]),
    "mdg_sqrt_psd_small": (_i32, [_ptr, _i64, _i64, _f64, _i32, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    "mdg_sqrt_psd_large_ws_bytes": (_sz, [_i64]),
    "mdg_sqrt_psd_large": (_i32, [_ptr, _i64, _i64, _f64, _i32, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    "mdg_rope_gather": (_i32, [_ptr, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _f64,
                                _ptr, _ptr]),
    "mdg_comm_unique_id": (_i32, [_ptr]),
    "mdg_comm_init": (_i32, [C.POINTER(_ptr), _i32, _i32, _ptr]),
    "mdg_allgather_layers": (_i32, [_ptr, _ptr, _sz, _ptr, _ptr]),
    "mdg_comm_destroy": (_i32, [_ptr]),
    "mdg_cast_transpose_f64_bf16": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _i64, _ptr]),
    "mdg_probe_mfma_f64": (_i32, [_i32, C.POINTER(_f64), _ptr]),
    "mdg_probe_mfma_i8": (_i32, [_i32, _i32, C.POINTER(_f64), _ptr]),
}

_lib = None


class ModeGPTLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the in-tree HIP library; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm must bring in ITS libamdhip64 first: the kernels run on torch's streams and device pointers, so
    # both have to share one HIP runtime instance (loading ours first binds /opt/rocm's copy and the process ends
    # up with a runtime that sees no device).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ModeGPTLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C modegpt_amd/csrc`). modegpt_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mdg_abi_version() != ABI_VERSION:
        raise ModeGPTLibraryError("libmodegpt_hip.so ABI version mismatch")
    _lib = lib
    # the library's cached device allocations are released while the HIP runtime is still alive (Python's atexit runs before
    # the C++ static destructors); the library itself makes no HIP call at exit
    import atexit
    atexit.register(lib.mdg_shutdown)
    return lib


def last_error() -> str:
    return load().mdg_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != MDG_OK:
        msg = last_error()
        if rc == MDG_ERR_NOT_PD:
            # same exception family torch.linalg.cholesky raises in the reference path
            import torch
            raise torch.linalg.LinAlgError(msg)
        raise RuntimeError(f"{what or 'modegpt_hip'} failed (status {rc}): {msg}")
