"""ctypes binding of libnrv_hip.so (the C ABI declared in include/nrv.h).

The library is the product: there is NO fallback.  If it cannot be loaded, or a call returns a
non-zero code, a RuntimeError is raised -- nothing here ever routes through PyTorch eager maths or
the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p, c_double

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "lib", "libnrv_hip.so")
LIB_PATH = _DEFAULT_LIB       # no environment override: what runs is the in-tree library (tools/_devlib.py swaps it for A/B runs)

NRV_F32, NRV_BF16, NRV_U8 = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, EPI_DGELU, EPI_BIAS_GELU_Q8, EPI_DGELU_Q8 = 0, 1, 2, 3, 4, 5, 6
PATCH_P1P2C, PATCH_CP1P2 = 0, 1
ABI_VERSION = 13
ATTN_QKV_BLOCKED, ATTN_OUT_BLOCKED = 1, 2      # include/nrv.h: NRV_ATTN_*_BLOCKED



class TnProblem(ctypes.Structure):
    """include/nrv.h `nrv_tn_problem`: one weight gradient of a grouped launch."""
    _fields_ = [("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64), ("C", c_void_p), ("ldc", c_int64),
                ("M", c_int64), ("N", c_int64), ("beta", c_float), ("dbias", c_void_p), ("dbias_beta", c_float)]


# name -> (restype, argtypes); every symbol include/nrv.h declares (tests/test_abi.py checks the two agree)
SIGNATURES = {
    "nrv_abi_version": (c_int, []),
    "nrv_error_string": (c_char_p, [c_int]),
    "nrv_layernorm_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_int, c_float, c_void_p]),
    "nrv_layernorm_bwd_workspace": (c_size_t, [c_int64, c_int]),
    "nrv_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_size_t, c_int64, c_int, c_void_p]),
    "nrv_gemm_nt_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64,
                                 c_int64, c_int64, c_int64, c_int, c_void_p,
                                 c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64,
                                 c_int64, c_int64, c_int64, c_void_p]),
    "nrv_gemm_tn_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "nrv_gemm_tn_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                 c_int64, c_int64, c_int64, c_float, c_int64, c_int64, c_int64,
                                 c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "nrv_gemm_tn_grouped_workspace": (c_size_t, [c_void_p, c_int, c_int64]),
    "nrv_gemm_tn_grouped_bf16": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_size_t, c_void_p]),
    "nrv_colsum_workspace": (c_size_t, [c_int64, c_int64]),
    "nrv_colsum_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_float, c_void_p, c_size_t, c_void_p]),
    "nrv_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "nrv_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "nrv_attn_probs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "nrv_attn_sinkhorn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "nrv_attn_sinkhorn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "nrv_patch_unfold": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "nrv_cast_transpose": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "nrv_cast_transpose_batched": (c_int, [c_void_p, c_int, c_int64, c_void_p]),
    "nrv_cast_f32_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nrv_dropout_add_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "nrv_mask_mul_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "nrv_mask_mul_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "nrv_gather_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "nrv_scatter_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "nrv_sumsq_workspace": (c_size_t, [c_int64]),
    "nrv_sumsq_f32": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrv_adamw_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64,
                              c_double, c_double, c_double, c_double, c_double, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "nrv_sinkhorn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "nrv_sinkhorn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "nrv_bgemm": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                          c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "nrv_set_reserved_cus": (c_int, [c_int]),
    "nrv_probe": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p]),
}

_lock = threading.Lock()
_lib = None


class NrvError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libnrv_hip.so once; raise loudly if it is missing (run `python -m noise_robust_vit_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH) and LIB_PATH == _DEFAULT_LIB:
            # the in-tree library is a build product (git-ignored): compile it if the toolchain is here.  This is
            # the same HIP code path, not a fallback; without hipcc the error below is raised.
            try:
                from . import build as _build
                _build.build()
            except Exception as e:        # noqa: BLE001 -- reported below
                raise NrvError(f"{LIB_PATH} is missing and building it failed ({e}); the HIP kernels are the product "
                               "and there is no fallback path") from e
        if not os.path.exists(LIB_PATH):
            raise NrvError(
                f"{LIB_PATH} not found: the HIP kernels are the product and there is no fallback path. "
                "Build them with `python -m noise_robust_vit_amd.build` (needs hipcc, gfx950 target).")
        _lib = bind(LIB_PATH)
    return _lib


def bind(path: str) -> ctypes.CDLL:
    """dlopen `path` and set the prototype of every symbol include/nrv.h declares."""
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.nrv_abi_version()
    if got != ABI_VERSION:
        raise NrvError(f"{path}: ABI version {got} != binding version {ABI_VERSION}; rebuild the library")
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().nrv_error_string(code)
        raise NrvError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")
