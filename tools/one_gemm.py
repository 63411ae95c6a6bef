#!/usr/bin/env python3
"""Run a handful of launches of the hot kernels on ViT-B/16 shapes (profiling target for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
T = 50432
which = sys.argv[1] if len(sys.argv) > 1 else "all"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def r(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * .5).to(dt)
if which in ("nt", "all"):
    A = r(T, 768); B = r(3072, 768); bias = r(3072, dt=torch.float32); u = torch.empty(T, 3072, dtype=torch.bfloat16, device=dev)
    for _ in range(n): K.gemm_nt(A, B, epilogue=EPI_BIAS_GELU, bias=bias, aux_out=u)
    A2 = r(T, 3072); B2 = r(768, 3072)
    for _ in range(n): K.gemm_nt(A2, B2)
if which in ("tn", "all"):
    A = r(T, 3072); B = r(T, 768)
    for _ in range(n): K.gemm_tn(A, B)
if which in ("attn", "all"):
    qkv = r(256 * 197, 3 * 12 * 64)
    for _ in range(n):
        o, lse = K.attn_fwd(qkv, 256, 197, 12, 64, 0.125)
        K.attn_bwd(qkv, o, o, lse, 256, 197, 12, 64, 0.125)
torch.cuda.synchronize()
