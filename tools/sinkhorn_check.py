#!/usr/bin/env python3
"""Sinkhorn backward of a developer build against the product build on the same inputs (and both against the fp32
definition through autograd for one small case).  GPU only; dev tool.   python tools/sinkhorn_check.py skf"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
lib = sys.argv[1]
dev = torch.device("cuda:0")
worst = 0.0
for (B, N, H) in [(2, 197, 3), (1, 49, 2), (1, 256, 1), (2, 5, 1), (1, 65, 1), (1, 100, 1), (1, 129, 2), (1, 177, 1), (1, 224, 1), (1, 196, 12), (3, 16, 2), (1, 33, 1)]:
    g = torch.Generator(device=dev).manual_seed(N)
    qkv = (torch.randn(B * N, 3 * H * 64, device=dev, generator=g) * 0.7).bfloat16()
    do = (torch.randn(B * N, H * 64, device=dev, generator=g) * 0.5).bfloat16()
    outs = {}
    for l in ("product", lib):
        _devlib.use_library(l)
        o, lse, scal = K.attn_sinkhorn_fwd(qkv, B, N, H, 64, 0.125)
        outs[l] = K.attn_sinkhorn_bwd(qkv, do, lse, scal, B, N, H, 64, 0.125).float()
    torch.cuda.synchronize()
    a, b = outs["product"], outs[lib]
    parts = {}
    for name, sl in (("dq", slice(0, H * 64)), ("dk", slice(H * 64, 2 * H * 64)), ("dv", slice(2 * H * 64, 3 * H * 64))):
        d = (a[:, sl] - b[:, sl]).abs().max().item() / max(a[:, sl].abs().max().item(), 1e-9)
        parts[name] = d
        worst = max(worst, d)
    print(f"B{B} N{N} H{H}: rel max diff dq {parts['dq']:.2e} dk {parts['dk']:.2e} dv {parts['dv']:.2e}  finite {bool(torch.isfinite(b).all())}", flush=True)
print("worst", worst)
