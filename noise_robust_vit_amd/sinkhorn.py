"""`robust=True` attention: softmax + Sinkhorn row/column normalisation (utils.py:1025-1037), fused on chip.

Kernels: csrc/nrv_sinkhorn.hip (C ABI nrv_attn_sinkhorn_fwd / _bwd).  `robust=True` never runs the softmax kernel.
"""
from __future__ import annotations

from . import kernels as K


def require_available() -> None:
    """Kept as the single switch the encoder consults before taking the robust path."""
    from . import _lib
    lib = _lib.load()
    if not hasattr(lib, "nrv_attn_sinkhorn_fwd"):
        raise _lib.NrvError("libnrv_hip.so was built without the Sinkhorn attention kernels")


def attn_fwd(qkv, B, N, H, dh, scale):
    out, lse, scal = K.attn_sinkhorn_fwd(qkv, B, N, H, dh, scale)
    return out, (lse, scal)


def attn_bwd(qkv, out, dout, aux, B, N, H, dh, scale):
    lse, scal = aux
    return K.attn_sinkhorn_bwd(qkv, dout, lse, scal, B, N, H, dh, scale)
