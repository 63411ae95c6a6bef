"""Tensor-level wrappers over the C ABI (include/nrv.h).  No autograd, no fallbacks.

PyTorch is used for device memory (caching allocator) and the current HIP stream only; every
arithmetic result comes from a kernel in libnrv_hip.so.  All tensors must live on a HIP device
(`tensor.is_cuda`), be contiguous in their last dimension and 16-byte aligned.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_Q8, EPI_BIAS_RESIDUAL, EPI_DGELU, EPI_DGELU_Q8, EPI_NONE, NRV_BF16, NRV_F32, NRV_U8,
                   PATCH_CP1P2, PATCH_P1P2C, NrvError, check)

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise NrvError(f"{name} must be on the MI355X (HIP) device; this path has no CPU implementation")


def _dt(t: Tensor, name: str) -> int:
    if t.dtype == torch.float32:
        return NRV_F32
    if t.dtype == torch.bfloat16:
        return NRV_BF16
    raise NrvError(f"{name}: dtype {t.dtype} not supported (fp32 or bf16)")


def _bf16(t: Tensor, name: str) -> None:
    _dev(t, name)
    if t.dtype != torch.bfloat16:
        raise NrvError(f"{name} must be bf16, got {t.dtype}")


def _f32(t: Tensor, name: str) -> None:
    _dev(t, name)
    if t.dtype != torch.float32:
        raise NrvError(f"{name} must be fp32, got {t.dtype}")


def _rows2d(t: Tensor, name: str) -> Tuple[int, int, int]:
    """(rows, cols, ld) of a 2-D view with unit stride in the last dim."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise NrvError(f"{name} must be 2-D with contiguous rows, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _workspace(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def set_reserved_cus(n: int) -> int:
    """CUs the GEMM launches leave free for a collective's kernels (include/nrv.h); returns the previous value."""
    prev = _lib.load().nrv_set_reserved_cus(int(n))
    if prev < 0:
        check(prev, "nrv_set_reserved_cus")
    return prev


# ----------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): HIP events on the stream the kernel runs on
# ----------------------------------------------------------------------------------------------
_PROF = None


class LaunchProfile:
    """Collects (kernel class, algorithmic FLOPs, algorithmic bytes, start event, end event) per C-ABI call."""

    def __init__(self) -> None:
        self.records = []

    def __enter__(self):
        global _PROF
        _PROF = self
        return self

    def __exit__(self, *exc):
        global _PROF
        _PROF = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, s, e in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


def _run(name: str, flops: float, nbytes: float, call, what: str) -> None:
    if _PROF is None:
        check(call(), what)
        return
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    code = call()
    e.record()
    check(code, what)
    _PROF.records.append((name, flops, nbytes, s, e))


# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float):
    """x [rows, dim] fp32|bf16 -> (y bf16, mean fp32, rstd fp32).  nn.LayerNorm (simple_vit.py:38,54; vit.py:104,115)."""
    _dev(x, "x"); _f32(gamma, "gamma"); _f32(beta, "beta")
    x = x.contiguous()
    rows, dim = x.shape
    y = torch.empty(rows, dim, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _run("layernorm_fwd", 0.0, rows * dim * (x.element_size() + 2),
         lambda: lib.nrv_layernorm_fwd(x.data_ptr(), _dt(x, "x"), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                       mean.data_ptr(), rstd.data_ptr(), rows, dim, float(eps), _stream()),
         "nrv_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dres: Optional[Tensor] = None,
                  want_f32: bool = True, want_bf16: bool = False,
                  dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None, accumulate: bool = False):
    """Returns (dx_f32|None, dx_bf16|None, dgamma, dbeta); dx = dres + LN'(dy)."""
    _bf16(dy, "dy"); _dev(x, "x"); _f32(gamma, "gamma")
    rows, dim = x.shape
    lib = _lib.load()
    dx32 = torch.empty(rows, dim, dtype=torch.float32, device=x.device) if want_f32 else None
    dx16 = torch.empty(rows, dim, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    if dgamma is None:
        dgamma = torch.empty(dim, dtype=torch.float32, device=x.device); accumulate = False
    if dbeta is None:
        dbeta = torch.empty(dim, dtype=torch.float32, device=x.device)
    wsb = lib.nrv_layernorm_bwd_workspace(rows, dim)
    ws = _workspace(wsb, x.device)
    nb = rows * dim * (2 + x.element_size() + (dres.element_size() if dres is not None else 0)
                       + (4 if want_f32 else 0) + (2 if want_bf16 else 0))
    _run("layernorm_bwd", 0.0, nb,
         lambda: lib.nrv_layernorm_bwd(dy.data_ptr(), x.data_ptr(), _dt(x, "x"), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       _ptr(dres), _dt(dres, "dres") if dres is not None else 0,
                                       _ptr(dx32), _ptr(dx16), dgamma.data_ptr(), dbeta.data_ptr(), int(bool(accumulate)),
                                       ws.data_ptr(), ws.numel(), rows, dim, _stream()),
         "nrv_layernorm_bwd")
    return dx32, dx16, dgamma, dbeta


def gemm_nt(A: Tensor, B: Tensor, *, out_dtype: torch.dtype = torch.bfloat16, epilogue: int = EPI_NONE,
            bias: Optional[Tensor] = None, aux: Optional[Tensor] = None, aux_row_mod: int = 0,
            aux_out: Optional[Tensor] = None, out: Optional[Tensor] = None,
            out_group: int = 0, out_group_stride: int = 0, out_row_offset: int = 0) -> Tensor:
    """C[M,N] = A[M,K] . B[N,K]^T with a fused epilogue (include/nrv.h nrv_gemm_nt_bf16)."""
    _bf16(A, "A"); _bf16(B, "B")
    M, K, lda = _rows2d(A, "A")
    N, K2, ldb = _rows2d(B, "B")
    if K != K2:
        raise NrvError(f"gemm_nt: K mismatch {K} vs {K2}")
    if out is None:
        if out_group:
            raise NrvError("gemm_nt: an output row remap needs a caller-provided `out`")
        out = torch.empty(M, N, dtype=out_dtype, device=A.device)
    _, _, ldc = _rows2d(out, "out")
    ld_aux = 0
    aux_code = 0
    if aux is not None:
        _dev(aux, "aux")
        if (epilogue == EPI_DGELU_Q8) != (aux.dtype == torch.uint8):
            raise NrvError("gemm_nt: the 8-bit gelu' stream (uint8) goes with EPI_DGELU_Q8 and with nothing else")
        aux_code = NRV_U8 if aux.dtype == torch.uint8 else _dt(aux, "aux")
        arows, acols, ld_aux = _rows2d(aux, "aux")
        # the epilogue reads aux through raw pointers: row m % aux_row_mod, or the (remapped) output row
        need = aux_row_mod if aux_row_mod else (M if not out_group else (M - 1) // out_group * out_group_stride + (M - 1) % out_group + out_row_offset + 1)
        if epilogue == EPI_DGELU_Q8:
            need = (need + 1) // 2 * 2          # the byte stream is stored in row pairs (include/nrv.h)
        if arows < need or acols < N:
            raise NrvError(f"gemm_nt: aux is [{arows}, {acols}] but the epilogue reads rows < {need}, columns < {N}")
    if out_group:
        need = (M - 1) // out_group * out_group_stride + (M - 1) % out_group + out_row_offset + 1
        if out.shape[0] < need:
            raise NrvError(f"gemm_nt: out has {out.shape[0]} rows, the row remap writes up to row {need - 1}")
    elif out.shape[0] < M or out.shape[1] < N:
        raise NrvError(f"gemm_nt: out is {tuple(out.shape)}, result is [{M}, {N}]")
    ld_ao = 0
    if aux_out is not None:
        if epilogue == EPI_BIAS_GELU_Q8:
            _dev(aux_out, "aux_out")
            if aux_out.dtype != torch.uint8 or aux_out.shape[0] < (M + 1) // 2 * 2 or N % 64:
                raise NrvError("gemm_nt: EPI_BIAS_GELU_Q8 writes the gelu' stream as uint8 in row pairs: M rounded up to even rows, N % 64 == 0")
        else:
            _bf16(aux_out, "aux_out")
        _, _, ld_ao = _rows2d(aux_out, "aux_out")
    if bias is not None:
        _f32(bias, "bias")
    lib = _lib.load()
    nb = 2 * (M * K + N * K) + M * N * out.element_size()
    if aux is not None and not aux_row_mod:
        nb += M * N * aux.element_size()
    if aux_out is not None:
        nb += M * N * aux_out.element_size()
    _run("gemm_nt", 2.0 * M * N * K, nb,
         lambda: lib.nrv_gemm_nt_bf16(A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), _dt(out, "out"), ldc,
                                      M, N, K, int(epilogue), _ptr(bias),
                                      _ptr(aux), aux_code, ld_aux, int(aux_row_mod),
                                      _ptr(aux_out), ld_ao, int(out_group), int(out_group_stride), int(out_row_offset),
                                      _stream()),
         "nrv_gemm_nt_bf16")
    return out


def gemm_tn(A: Tensor, B: Tensor, *, out: Optional[Tensor] = None, beta: float = 0.0,
            a_group: int = 0, a_group_stride: int = 0, a_row_offset: int = 0, T: Optional[int] = None,
            dbias: Optional[Tensor] = None, dbias_beta: float = 0.0, want_dbias: bool = False):
    """C[M,N] fp32 = beta*C + sum_t A[t,M] B[t,N]  (weight gradient).  `T` limits the token rows used (default B.shape[0]).

    With `want_dbias` (or a `dbias` output) the bias gradient sum_t A[t,:] is produced by the same kernel and the
    call returns (C, dbias)."""
    _bf16(A, "A"); _bf16(B, "B")
    Ta, M, lda = _rows2d(A, "A")
    Tb, N, ldb = _rows2d(B, "B")
    T = Tb if T is None else T
    if a_group == 0 and Ta != Tb:
        raise NrvError(f"gemm_tn: token count mismatch {Ta} vs {Tb}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
        beta = 0.0
    _f32(out, "out")
    _, _, ldc = _rows2d(out, "out")
    if want_dbias and dbias is None:
        dbias = torch.empty(M, dtype=torch.float32, device=A.device)
        dbias_beta = 0.0
    if dbias is not None:
        _f32(dbias, "dbias")
    lib = _lib.load()
    ws = _workspace(lib.nrv_gemm_tn_workspace(M, N, T), A.device)
    _run("gemm_tn", 2.0 * M * N * T, 2 * T * (M + N) + 4 * M * N,
         lambda: lib.nrv_gemm_tn_bf16(A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, T, float(beta),
                                      int(a_group), int(a_group_stride), int(a_row_offset),
                                      _ptr(dbias), float(dbias_beta), ws.data_ptr(), ws.numel(), _stream()),
         "nrv_gemm_tn_bf16")
    if dbias is not None:
        return out, dbias
    return out


def gemm_tn_grouped(problems) -> list:
    """ALL weight gradients of a layer in one stream-K launch (include/nrv.h nrv_gemm_tn_grouped_bf16).  `problems`: up to 4
    dicts {A: dy bf16 [T, M], B: x bf16 [T, N], out: fp32 [M, N] | None, beta, dbias: fp32 [M] | None | True (allocate),
    dbias_beta}; all with the same T.  Returns [(out, dbias | None)].  Groups the kernel does not take (tiny T) run as
    separate `gemm_tn` calls -- the same arithmetic per gradient, another summation order."""
    if not problems:
        return []
    lib = _lib.load()
    T = problems[0]["A"].shape[0]
    arr = (_lib.TnProblem * len(problems))()
    outs = []
    for i, q in enumerate(problems):
        A, B = q["A"], q["B"]
        _bf16(A, "A"); _bf16(B, "B")
        Ta, M, lda = _rows2d(A, "A")
        Tb, N, ldb = _rows2d(B, "B")
        if Ta != T or Tb != T:
            raise NrvError(f"gemm_tn_grouped: problem {i} has {Ta} / {Tb} token rows, the group has {T}")
        out, beta = q.get("out"), float(q.get("beta", 0.0))
        if out is None:
            out, beta = torch.empty(M, N, dtype=torch.float32, device=A.device), 0.0
        _f32(out, "out")
        if out.shape[0] < M or out.shape[1] < N:
            raise NrvError(f"gemm_tn_grouped: out of problem {i} is {tuple(out.shape)}, the gradient is [{M}, {N}]")
        _, _, ldc = _rows2d(out, "out")
        dbias, dbb = q.get("dbias"), float(q.get("dbias_beta", 0.0))
        if dbias is True:
            dbias, dbb = torch.empty(M, dtype=torch.float32, device=A.device), 0.0
        if dbias is not None:
            _f32(dbias, "dbias")
            if dbias.numel() < M:
                raise NrvError("gemm_tn_grouped: dbias is shorter than the gradient's rows")
        arr[i] = _lib.TnProblem(A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, beta, _ptr(dbias), dbb)
        outs.append((out, dbias))
    nbytes = int(lib.nrv_gemm_tn_grouped_workspace(ctypes.addressof(arr), len(problems), T)) if len(problems) <= 4 else 0
    if nbytes == 0:                                   # not taken by the grouped kernel: one split-K launch per gradient
        for i, (q, (out, dbias)) in enumerate(zip(problems, outs)):
            given_db = q.get("dbias") is not None and q.get("dbias") is not True
            gemm_tn(q["A"], q["B"], out=out, beta=float(arr[i].beta), dbias=dbias,
                    dbias_beta=float(q.get("dbias_beta", 0.0)) if given_db else 0.0)
        return outs
    ws = _workspace(nbytes, problems[0]["A"].device)
    flops = sum(2.0 * T * q["A"].shape[1] * q["B"].shape[1] for q in problems)
    nb = sum(2 * T * (q["A"].shape[1] + q["B"].shape[1]) + 4 * q["A"].shape[1] * q["B"].shape[1] for q in problems)
    _run("gemm_tn", flops, nb,
         lambda: lib.nrv_gemm_tn_grouped_bf16(ctypes.addressof(arr), len(problems), T, ws.data_ptr(), ws.numel(), _stream()),
         "nrv_gemm_tn_grouped_bf16")
    return outs


def colsum(X: Tensor, *, out: Optional[Tensor] = None, beta: float = 0.0) -> Tensor:
    """out[n] = beta*out[n] + sum_t X[t,n]  (bias gradient)."""
    _bf16(X, "X")
    T, N, ld = _rows2d(X, "X")
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
        beta = 0.0
    lib = _lib.load()
    ws = _workspace(lib.nrv_colsum_workspace(T, N), X.device)
    _run("colsum", 0.0, 2 * T * N,
         lambda: lib.nrv_colsum_bf16(X.data_ptr(), ld, out.data_ptr(), T, N, float(beta), ws.data_ptr(), ws.numel(), _stream()),
         "nrv_colsum_bf16")
    return out


def attn_fwd(qkv: Tensor, B: int, N: int, H: int, dh: int, scale: float, layout: int = 0):
    """qkv bf16 [B*N, 3*H*dh] -> (out bf16 [B*N, H*dh], lse fp32 [B,H,N]).  simple_vit.py:68-75.
    `layout` (include/nrv.h NRV_ATTN_*_BLOCKED): qkv given as [3*H, B*N, dh] / out returned as [H, B*N, dh]."""
    _bf16(qkv, "qkv")
    if not qkv.is_contiguous() or qkv.numel() != B * N * 3 * H * dh:
        raise NrvError("attn_fwd: qkv must be contiguous [B*N, 3*H*dh] (or [3*H, B*N, dh] with the blocked layout)")
    out = (torch.empty(H, B * N, dh, dtype=torch.bfloat16, device=qkv.device) if layout & _lib.ATTN_OUT_BLOCKED
           else torch.empty(B * N, H * dh, dtype=torch.bfloat16, device=qkv.device))
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    _run("attn_fwd", 4.0 * B * H * N * N * dh, 2 * B * N * H * dh * 4,
         lambda: lib.nrv_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, dh, float(scale), int(layout), _stream()),
         "nrv_attn_fwd")
    return out, lse


def attn_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, B: int, N: int, H: int, dh: int, scale: float,
             layout: int = 0) -> Tensor:
    _bf16(qkv, "qkv"); _bf16(out, "out"); _bf16(dout, "dout"); _f32(lse, "lse")
    if not (qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous()):
        raise NrvError("attn_bwd: operands must be contiguous")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * H * N, dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    _run("attn_bwd", 10.0 * B * H * N * N * dh, 2 * B * N * H * dh * 8,
         lambda: lib.nrv_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                  delta.data_ptr(), B, N, H, dh, float(scale), int(layout), _stream()),
         "nrv_attn_bwd")
    return dqkv


def attn_probs(qkv: Tensor, lse: Tensor, B: int, N: int, H: int, dh: int, scale: float, layout: int = 0) -> Tensor:
    """fp32 [B, H, N, N] softmax probabilities recomputed from q, k and the saved log-sum-exp (introspection only)."""
    _bf16(qkv, "qkv"); _f32(lse, "lse")
    probs = torch.empty(B, H, N, N, dtype=torch.float32, device=qkv.device)
    check(_lib.load().nrv_attn_probs(qkv.data_ptr(), lse.data_ptr(), probs.data_ptr(), B, N, H, dh, float(scale), int(layout), _stream()),
          "nrv_attn_probs")
    return probs


def _sinkhorn_fused_shape(N: int, dh: int) -> bool:
    """Shapes the fused kernels hold on chip (csrc/nrv_sinkhorn.hip): the head's whole [N, N] matrix in registers."""
    return dh == 64 and N <= 256


# composed robust attention: the forward's normalised matrix P7 [B,H,N,N] fp32 is handed to the backward (which otherwise recomputes it:
# the scores GEMM + 5 passes over the matrix) when it is at most this many bytes per layer -- 288 GB of HBM are there to be used
SINKHORN_KEEP_P_BYTES = 2 << 30


def attn_sinkhorn_fwd(qkv: Tensor, B: int, N: int, H: int, dh: int, scale: float, saved: Optional[dict] = None):
    """robust=True attention (utils.py:1025-1037): returns (out bf16, lse fp32 [B,H,N], scalings fp32 [B,H,7,N]).
    N <= 256 and dh == 64: the fused kernel.  Any other shape (vit_h_14, 384-px checkpoints, other head dims): composed from
    the batched GEMM and the stand-alone Sinkhorn op on materialised [B,H,N,N] scores -- the reference's own structure.
    `saved` (a dict the caller keeps for the backward and passes to attn_sinkhorn_bwd): the composed path leaves P7 in it."""
    _bf16(qkv, "qkv")
    if not qkv.is_contiguous() or qkv.numel() != B * N * 3 * H * dh:
        raise NrvError("attn_sinkhorn_fwd: qkv must be contiguous [B*N, 3*H*dh]")
    if not _sinkhorn_fused_shape(N, dh):
        return _attn_sinkhorn_fwd_composed(qkv, B, N, H, dh, scale, saved)
    out = torch.empty(B * N, H * dh, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    scal = torch.empty(B, H, 7, N, dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    _run("attn_sinkhorn_fwd", 4.0 * B * H * N * N * dh, 2 * B * N * H * dh * 4,
         lambda: lib.nrv_attn_sinkhorn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), scal.data_ptr(),
                                           B, N, H, dh, float(scale), _stream()),
         "nrv_attn_sinkhorn_fwd")
    return out, lse, scal


def attn_sinkhorn_bwd(qkv: Tensor, dout: Tensor, lse: Tensor, scal: Tensor, B: int, N: int, H: int, dh: int, scale: float,
                      saved: Optional[dict] = None) -> Tensor:
    _bf16(qkv, "qkv"); _bf16(dout, "dout"); _f32(lse, "lse"); _f32(scal, "scal")
    if not (qkv.is_contiguous() and dout.is_contiguous()):
        raise NrvError("attn_sinkhorn_bwd: operands must be contiguous")
    if not _sinkhorn_fused_shape(N, dh):
        return _attn_sinkhorn_bwd_composed(qkv, dout, lse, scal, B, N, H, dh, scale, saved)
    dqkv = torch.empty_like(qkv)
    lib = _lib.load()
    _run("attn_sinkhorn_bwd", 10.0 * B * H * N * N * dh, 2 * B * N * H * dh * 7,
         lambda: lib.nrv_attn_sinkhorn_bwd(qkv.data_ptr(), dout.data_ptr(), lse.data_ptr(), scal.data_ptr(), dqkv.data_ptr(),
                                           B, N, H, dh, float(scale), _stream()),
         "nrv_attn_sinkhorn_bwd")
    return dqkv


# ---- batched strided GEMM + the composed robust attention -----------------------------------------------------------------
def bgemm(A, a_str, B_, b_str, C, c_str, G1: int, G2: int, M: int, N: int, K: int, alpha: float = 1.0) -> None:
    """C[g1,g2] = alpha * A[g1,g2] . B[g1,g2] (include/nrv.h nrv_bgemm).  A, B, C: (tensor, element offset) pairs; *_str =
    (row stride, column stride, g1 stride, g2 stride) in elements.  The tensors are only memory: the strides do the addressing,
    and the caller guarantees that every addressed element lies inside its tensor (checked below)."""
    lib = _lib.load()
    ptrs = []
    for (t, off), st, rows, cols, name in ((A, a_str, M, K, "A"), (B_, b_str, K, N, "B"), (C, c_str, M, N, "C")):
        _dev(t, name)
        last = off + (rows - 1) * st[0] + (cols - 1) * st[1] + (G1 - 1) * st[2] + (G2 - 1) * st[3]
        if off < 0 or min(st) < 0 or last >= t.numel():
            raise NrvError(f"bgemm: operand {name} addresses element {last} of a tensor with {t.numel()}")
        ptrs.append((t.data_ptr() + off * t.element_size(), _dt(t, name)))
    _run("bgemm", 2.0 * G1 * G2 * M * N * K, 0.0,
         lambda: lib.nrv_bgemm(ptrs[0][0], ptrs[0][1], *[int(v) for v in a_str], ptrs[1][0], ptrs[1][1], *[int(v) for v in b_str],
                               ptrs[2][0], ptrs[2][1], *[int(v) for v in c_str], G1, G2, M, N, K, float(alpha), _stream()),
         "nrv_bgemm")


def _head_strides(H: int, dh: int, width: int, N: int):
    """(row, col, batch, head) element strides of one head's [N, dh] slice inside a [B*N, width] projection."""
    return (width, 1, N * width, dh)


def _sinkhorn_scores(qkv: Tensor, B: int, N: int, H: int, dh: int, scale: float) -> Tensor:
    """S[b,h] = scale * q k^T, fp32 [B,H,N,N] (simple_vit.py:70)."""
    W = 3 * H * dh
    rs, cs, bs, hs = _head_strides(H, dh, W, N)
    S = torch.empty(B, H, N, N, dtype=torch.float32, device=qkv.device)
    bgemm((qkv, 0), (rs, cs, bs, hs), (qkv, H * dh), (cs, rs, bs, hs), (S, 0), (N, 1, H * N * N, N * N), B, H, N, N, dh, scale)
    return S


def _attn_sinkhorn_fwd_composed(qkv: Tensor, B: int, N: int, H: int, dh: int, scale: float, saved: Optional[dict] = None):
    W = 3 * H * dh
    rs, cs, bs, hs = _head_strides(H, dh, W, N)
    S = _sinkhorn_scores(qkv, B, N, H, dh, scale)
    P, lse, avec, bvec = sinkhorn_fwd(S, iters=3)                 # SinkhornAttention on the materialised scores (utils.py:1031-1037)
    del S
    out = torch.empty(B * N, H * dh, dtype=torch.bfloat16, device=qkv.device)
    ors, ocs, obs, ohs = _head_strides(H, dh, H * dh, N)
    # O[b,h] = P7 v   (attn v, simple_vit.py:74; P7 enters the product in bf16 as in the fused kernels)
    bgemm((P, 0), (N, 1, H * N * N, N * N), (qkv, 2 * H * dh), (rs, cs, bs, hs), (out, 0), (ors, ocs, obs, ohs), B, H, N, dh, N, 1.0)
    # the model keeps ONE form of the saved statistics: [B,H,7,N] = a1 b1 a2 b2 a3 b3 a4 (cumulative), as the fused kernel writes it
    scal = torch.empty(B, H, 7, N, dtype=torch.float32, device=qkv.device)
    scal[:, :, 0::2] = avec.reshape(B, H, 4, N)
    scal[:, :, 1::2] = bvec.reshape(B, H, 3, N)
    if saved is not None and P.numel() * 4 <= SINKHORN_KEEP_P_BYTES:
        saved["P7"] = P
    return out, lse.reshape(B, H, N), scal


def _attn_sinkhorn_bwd_composed(qkv: Tensor, dout: Tensor, lse: Tensor, scal: Tensor, B: int, N: int, H: int, dh: int, scale: float,
                                saved: Optional[dict] = None) -> Tensor:
    W = 3 * H * dh
    rs, cs, bs, hs = _head_strides(H, dh, W, N)
    ors, ocs, obs, ohs = _head_strides(H, dh, H * dh, N)
    mat = (N, 1, H * N * N, N * N)                                 # a [B,H,N,N] fp32 matrix
    matT = (1, N, H * N * N, N * N)                                # ... read transposed
    S = _sinkhorn_scores(qkv, B, N, H, dh, scale)                  # recomputed (one GEMM): the Sinkhorn backward rebuilds P0 from it
    P = saved.pop("P7", None) if saved is not None else None       # the forward's matrix when it was small enough to keep
    if P is None:
        P, _, _, _ = sinkhorn_fwd(S, iters=3)
    avec = scal[:, :, 0::2].reshape(B * H, 4, N).contiguous()
    bvec = scal[:, :, 1::2].reshape(B * H, 3, N).contiguous()
    dqkv = torch.empty_like(qkv)
    # dV = P7^T dO
    bgemm((P, 0), matT, (dout, 0), (ors, ocs, obs, ohs), (dqkv, 2 * H * dh), (rs, cs, bs, hs), B, H, N, dh, N, 1.0)
    # dP7 = dO v^T, reusing P's storage would alias an operand of nothing that follows: a buffer of its own keeps it simple
    dP = torch.empty_like(P)
    bgemm((dout, 0), (ors, ocs, obs, ohs), (qkv, 2 * H * dh), (cs, rs, bs, hs), (dP, 0), mat, B, H, N, N, dh, 1.0)
    del P
    dS = sinkhorn_bwd(S, dP, lse.reshape(B * H, N).contiguous(), avec, bvec, iters=3)
    del dP, S
    # dQ = scale dS k ;  dK = scale dS^T q
    bgemm((dS, 0), mat, (qkv, H * dh), (rs, cs, bs, hs), (dqkv, 0), (rs, cs, bs, hs), B, H, N, dh, N, scale)
    bgemm((dS, 0), matT, (qkv, 0), (rs, cs, bs, hs), (dqkv, H * dh), (rs, cs, bs, hs), B, H, N, dh, N, scale)
    return dqkv


def patch_unfold(img: Tensor, p: int, layout: int) -> Tensor:
    """img [B,C,H,W] fp32|bf16 -> patches bf16 [B*(H/p)*(W/p), FP]; FP = C*p*p rounded up to a multiple of 8, the extra
    columns are zero (only patch sizes like 14 have any)."""
    _dev(img, "img")
    img = img.contiguous()
    B, C, H, W = img.shape
    out = torch.empty(B * (H // p) * (W // p), (C * p * p + 7) // 8 * 8, dtype=torch.bfloat16, device=img.device)
    lib = _lib.load()
    _run("patch_unfold", 0.0, img.numel() * (img.element_size() + 2),
         lambda: lib.nrv_patch_unfold(img.data_ptr(), _dt(img, "img"), out.data_ptr(), B, C, H, W, p, layout, _stream()),
         "nrv_patch_unfold")
    return out


def cast_transpose(w: Tensor, need_t: bool = True):
    """w fp32 [R,C] -> (w_bf16 [R,C], wT_bf16 [C,R] | None)."""
    _f32(w, "w")
    w = w.contiguous()
    R, C = w.shape
    wb = torch.empty(R, C, dtype=torch.bfloat16, device=w.device)
    wt = torch.empty(C, R, dtype=torch.bfloat16, device=w.device) if need_t else None
    lib = _lib.load()
    _run("cast_transpose", 0.0, R * C * 8,
         lambda: lib.nrv_cast_transpose(w.data_ptr(), wb.data_ptr(), _ptr(wt), R, C, _stream()), "nrv_cast_transpose")
    return wb, wt


def cast_bf16(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    _f32(x, "x")
    x = x.contiguous()
    if out is None:
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    else:
        _bf16(out, "out")
        if out.numel() != x.numel() or not out.is_contiguous():
            raise NrvError("cast_bf16: out must be contiguous bf16 of the same size")
        y = out
    lib = _lib.load()
    _run("cast_bf16", 0.0, x.numel() * 6,
         lambda: lib.nrv_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "nrv_cast_f32_bf16")
    return y


def _keep_mask(keep: Tensor, like: Tensor, what: str) -> None:
    _dev(keep, "keep")
    if keep.dtype != torch.uint8 or not keep.is_contiguous() or keep.numel() != like.numel():
        raise NrvError(f"{what}: keep must be a contiguous uint8 mask with one byte per element")
    if like.numel() % 8:
        raise NrvError(f"{what}: the element count must be a multiple of 8")


def dropout_add(x: Tensor, y: Tensor, keep: Tensor, scale: float, out: Optional[Tensor] = None) -> Tensor:
    """out = x + y * (keep ? scale : 0): the fp32 residual stream takes a dropped branch output (include/nrv.h nrv_dropout_add_f32)."""
    _f32(x, "x"); _f32(y, "y")
    if not (x.is_contiguous() and y.is_contiguous()) or x.shape != y.shape:
        raise NrvError("dropout_add: x and y must be contiguous fp32 tensors of one shape")
    _keep_mask(keep, x, "dropout_add")
    o = torch.empty_like(x) if out is None else out
    lib = _lib.load()
    _run("dropout", 0.0, x.numel() * 13,
         lambda: lib.nrv_dropout_add_f32(x.data_ptr(), y.data_ptr(), keep.data_ptr(), o.data_ptr(), float(scale), x.numel(), _stream()),
         "nrv_dropout_add_f32")
    return o


def mask_mul(a: Tensor, keep: Tensor, scale: float, out: Optional[Tensor] = None) -> Tensor:
    """out = a * (keep ? scale : 0) on bf16 tensors (include/nrv.h nrv_mask_mul_bf16); `out=a` works in place."""
    _bf16(a, "a")
    if not a.is_contiguous():
        raise NrvError("mask_mul: a must be contiguous")
    _keep_mask(keep, a, "mask_mul")
    o = torch.empty_like(a) if out is None else out
    lib = _lib.load()
    _run("dropout", 0.0, a.numel() * 5,
         lambda: lib.nrv_mask_mul_bf16(a.data_ptr(), keep.data_ptr(), o.data_ptr(), float(scale), a.numel(), _stream()),
         "nrv_mask_mul_bf16")
    return o


def mask_mul_f32(a: Tensor, keep: Tensor, scale: float, out: Optional[Tensor] = None) -> Tensor:
    """out = a * (keep ? scale : 0) on fp32 tensors of any size (include/nrv.h nrv_mask_mul_f32)."""
    _f32(a, "a"); _dev(keep, "keep")
    if not a.is_contiguous() or keep.dtype != torch.uint8 or not keep.is_contiguous() or keep.numel() != a.numel():
        raise NrvError("mask_mul_f32: a contiguous fp32, keep a contiguous uint8 mask with one byte per element")
    o = torch.empty_like(a) if out is None else out
    lib = _lib.load()
    _run("dropout", 0.0, a.numel() * 9,
         lambda: lib.nrv_mask_mul_f32(a.data_ptr(), keep.data_ptr(), o.data_ptr(), float(scale), a.numel(), _stream()), "nrv_mask_mul_f32")
    return o


def attn_dropout_fwd(qkv: Tensor, B: int, N: int, H: int, dh: int, scale: float, robust: bool, keep: Optional[Tensor], pscale: float,
                     bias: Optional[Tensor] = None):
    """Attention with what the fused kernels do not take, composed on the materialised matrix like the robust attention beyond the
    fused shapes: an additive score bias (attention / key-padding masks of torch's MultiheadAttention: utils.py:741-751; -inf = masked)
    and / or dropout ON THE ATTENTION WEIGHTS (attention_dropout > 0: vit.py:108) -- scores (+ bias), softmax (the Sinkhorn op with
    0 iterations) or the Sinkhorn normalisation (robust), the keep mask [B,H,N,N], P v.  Returns (out bf16, saved) with what the
    backward needs (the matrix that met v, the statistics)."""
    _bf16(qkv, "qkv")
    W = 3 * H * dh
    rs, cs, bs, hs = _head_strides(H, dh, W, N)
    iters = 3 if robust else 0
    S = _sinkhorn_scores(qkv, B, N, H, dh, scale)
    if bias is not None:
        _f32(bias, "bias")
        S += bias                                   # broadcast over batch / heads: plumbing, once per call (not a BASELINE path)
    P, lse, avec, bvec = sinkhorn_fwd(S, iters=iters)
    del S
    if keep is not None:
        if tuple(keep.shape) != (B, H, N, N):
            raise NrvError(f"attn_dropout_fwd: keep must be a uint8 mask of shape {(B, H, N, N)}")
        mask_mul_f32(P, keep, pscale, out=P)
    out = torch.empty(B * N, H * dh, dtype=torch.bfloat16, device=qkv.device)
    ors, ocs, obs, ohs = _head_strides(H, dh, H * dh, N)
    bgemm((P, 0), (N, 1, H * N * N, N * N), (qkv, 2 * H * dh), (rs, cs, bs, hs), (out, 0), (ors, ocs, obs, ohs), B, H, N, dh, N, 1.0)
    return out, (P, lse, avec, bvec, keep, pscale, iters, bias)


def attn_dropout_bwd(qkv: Tensor, dout: Tensor, saved, B: int, N: int, H: int, dh: int, scale: float) -> Tensor:
    Pd, lse, avec, bvec, keep, pscale, iters, bias = saved
    W = 3 * H * dh
    rs, cs, bs, hs = _head_strides(H, dh, W, N)
    ors, ocs, obs, ohs = _head_strides(H, dh, H * dh, N)
    mat, matT = (N, 1, H * N * N, N * N), (1, N, H * N * N, N * N)
    dqkv = torch.empty_like(qkv)
    bgemm((Pd, 0), matT, (dout, 0), (ors, ocs, obs, ohs), (dqkv, 2 * H * dh), (rs, cs, bs, hs), B, H, N, dh, N, 1.0)          # dV = Pd^T dO
    dP = torch.empty_like(Pd)
    bgemm((dout, 0), (ors, ocs, obs, ohs), (qkv, 2 * H * dh), (cs, rs, bs, hs), (dP, 0), mat, B, H, N, N, dh, 1.0)           # d(Pd) = dO v^T
    if keep is not None:
        mask_mul_f32(dP, keep, pscale, out=dP)                                                                                  # dP
    S = _sinkhorn_scores(qkv, B, N, H, dh, scale)
    if bias is not None:
        S += bias                                   # masked scores are -inf: P0 = 0 there, and so is dS
    dS = sinkhorn_bwd(S, dP, lse, avec, bvec, iters=iters)
    del dP, S
    bgemm((dS, 0), mat, (qkv, H * dh), (rs, cs, bs, hs), (dqkv, 0), (rs, cs, bs, hs), B, H, N, dh, N, scale)
    bgemm((dS, 0), matT, (qkv, 0), (rs, cs, bs, hs), (dqkv, H * dh), (rs, cs, bs, hs), B, H, N, dh, N, scale)
    return dqkv


def gather_rows(src: Tensor, index: Tensor) -> Tensor:
    """out[r] = src[index[r]]; src fp32 [R, dim], index int64 [rows_out]."""
    _f32(src, "src"); _dev(index, "index")
    src = src.contiguous(); index = index.contiguous()
    out = torch.empty(index.numel(), src.shape[1], dtype=torch.float32, device=src.device)
    if index.dtype != torch.int64:
        raise NrvError(f"gather_rows: index must be int64, got {index.dtype}")
    check(_lib.load().nrv_gather_rows_f32(src.data_ptr(), index.data_ptr(), out.data_ptr(), index.numel(), src.shape[0], src.shape[1], _stream()),
          "nrv_gather_rows_f32")
    return out


def scatter_rows(dout: Tensor, index: Tensor, rows_src: int) -> Tensor:
    """dsrc = zeros[rows_src, dim]; dsrc[index[r]] = dout[r]  (indices unique)."""
    _f32(dout, "dout"); _dev(index, "index")
    dout = dout.contiguous(); index = index.contiguous()
    dsrc = torch.zeros(rows_src, dout.shape[1], dtype=torch.float32, device=dout.device)
    if index.dtype != torch.int64 or index.numel() != dout.shape[0]:
        raise NrvError("scatter_rows: index must be int64 with one entry per row of dout")
    check(_lib.load().nrv_scatter_rows_f32(dout.data_ptr(), index.data_ptr(), dsrc.data_ptr(), index.numel(), rows_src, dout.shape[1], _stream()),
          "nrv_scatter_rows_f32")
    return dsrc


def probe(which: int, data: Tensor, n_out: int) -> Tensor:
    _dev(data, "data")
    out = torch.zeros(n_out, dtype=torch.float32, device=data.device)
    check(_lib.load().nrv_probe(which, data.data_ptr(), out.data_ptr(), data.numel() * data.element_size(), _stream()), "nrv_probe")
    return out


def sumsq_workspace(n: int) -> int:
    return int(_lib.load().nrv_sumsq_workspace(int(n)))


def sumsq(x: Tensor, out: Tensor, ws: Tensor) -> None:
    """out[0] = sum(x^2) for a flat fp32 or bf16 buffer (deterministic); ws: fp32 scratch of nrv_sumsq_workspace bytes."""
    _dev(x, "x")
    lib = _lib.load()
    _run("optimizer", 0.0, x.numel() * x.element_size(),
         lambda: lib.nrv_sumsq_f32(x.data_ptr(), _dt(x, "x"), x.numel(), out.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size(), _stream()),
         "nrv_sumsq_f32")


def adamw_flat(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, step: int, gnorm_sq: Optional[Tensor], max_norm: float,
               step_scalars: Optional[Tensor] = None) -> None:
    """In-place clip + AdamW on flat buffers (include/nrv.h: nrv_adamw_f32): p, m, v fp32; g fp32 or bf16 (the reduced slabs of a
    bf16 gradient exchange, read in place).  `step_scalars`: device tensor of 3 floats that replaces the step-dependent scalars
    (graph replay)."""
    for t, n in ((p, "p"), (m, "m"), (v, "v")):
        _f32(t, n)
    _dev(g, "g")
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise NrvError("adamw_flat: buffers differ in length")
    lib = _lib.load()
    _run("optimizer", 0.0, p.numel() * (24 + g.element_size()),
         lambda: lib.nrv_adamw_f32(p.data_ptr(), g.data_ptr(), _dt(g, "g"), m.data_ptr(), v.data_ptr(), p.numel(),
                                   float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                   _ptr(gnorm_sq), float(max_norm), _ptr(step_scalars), _stream()),
         "nrv_adamw_f32")


def sinkhorn_fwd(scores: Tensor, iters: int = 3):
    """SinkhornAttention(scores) on a materialised fp32 score tensor [..., R, C] (utils.py:1025-1037) -> (P, lse, avec, bvec)."""
    _f32(scores, "scores")
    if scores.dim() < 2:
        raise NrvError("sinkhorn_fwd: scores must have at least two dimensions")
    s = scores.contiguous()
    R, C = s.shape[-2], s.shape[-1]
    G = s.numel() // (R * C)
    out = torch.empty_like(s)
    lse = torch.empty(G, R, dtype=torch.float32, device=s.device)
    avec = torch.empty(G, iters + 1, R, dtype=torch.float32, device=s.device)
    bvec = torch.empty(G, max(iters, 1), C, dtype=torch.float32, device=s.device)
    lib = _lib.load()
    _run("sinkhorn_norm_fwd", 0.0, s.numel() * 4 * (2 * iters + 3),
         lambda: lib.nrv_sinkhorn_fwd(s.data_ptr(), out.data_ptr(), lse.data_ptr(), avec.data_ptr(), bvec.data_ptr(),
                                      G, R, C, int(iters), _stream()), "nrv_sinkhorn_fwd")
    return out, lse, avec, bvec


def sinkhorn_bwd(scores: Tensor, dout: Tensor, lse: Tensor, avec: Tensor, bvec: Tensor, iters: int = 3) -> Tensor:
    _f32(scores, "scores"); _f32(dout, "dout")
    s = scores.contiguous()
    d = dout.contiguous()
    R, C = s.shape[-2], s.shape[-1]
    G = s.numel() // (R * C)
    ds = torch.empty_like(s)
    lib = _lib.load()
    _run("sinkhorn_norm_bwd", 0.0, s.numel() * 4 * (4 * iters + 6),
         lambda: lib.nrv_sinkhorn_bwd(s.data_ptr(), d.data_ptr(), lse.data_ptr(), avec.data_ptr(), bvec.data_ptr(), ds.data_ptr(),
                                      G, R, C, int(iters), _stream()), "nrv_sinkhorn_bwd")
    return ds


def cast_transpose_batched(jobs) -> None:
    """Re-stage many weights in ONE launch.  jobs: list of (w fp32 [R,C], wb bf16 [R,C], wt bf16 [C,R] | None), all on one device;
    returns a reusable handle via `build_cast_jobs`."""
    handle = build_cast_jobs(jobs)
    run_cast_jobs(handle)


def build_cast_jobs(jobs):
    """Device-side job table of include/nrv.h `nrv_cast_job` (7 x int64 per entry).  The tensors must stay alive (and in place)
    for as long as the handle is used."""
    import numpy as np
    if not jobs:
        return None
    rows, start = [], 0
    for w, wb, wt in jobs:
        _f32(w, "w"); _bf16(wb, "wb")
        if not (w.is_contiguous() and wb.is_contiguous() and (wt is None or wt.is_contiguous())):
            raise NrvError("cast jobs need contiguous tensors")
        R, C = w.shape
        tiles_c = (C + 63) // 64
        rows.append((w.data_ptr(), wb.data_ptr(), 0 if wt is None else wt.data_ptr(), R, C, start, tiles_c))
        start += ((R + 63) // 64) * tiles_c
    table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(jobs[0][0].device)
    return table, len(rows), start


def run_cast_jobs(handle) -> None:
    if handle is None:
        return
    table, n, total = handle
    lib = _lib.load()
    _run("cast_transpose", 0.0, 0.0, lambda: lib.nrv_cast_transpose_batched(table.data_ptr(), n, total, _stream()),
         "nrv_cast_transpose_batched")
