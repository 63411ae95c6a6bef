"""`robust=True` attention: softmax + Sinkhorn row/column normalisation (utils.py:1025-1037), fused on chip.

SURVEY.md §8f rank 1.  The kernels live in csrc/nrv_sinkhorn.hip; until that file is built into
libnrv_hip.so this module refuses loudly -- `robust=True` never silently runs the softmax kernel.
"""
from __future__ import annotations

from ._lib import NrvError


def require_available() -> None:
    raise NrvError("robust=True (Sinkhorn attention) is not built into libnrv_hip.so yet; "
                   "the softmax path is not substituted for it")


def attn_fwd(qkv, B, N, H, dh, scale):
    require_available()


def attn_bwd(qkv, out, dout, aux, B, N, H, dh, scale):
    require_available()
