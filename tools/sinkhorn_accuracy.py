#!/usr/bin/env python3
"""Gradient accuracy of the fused Sinkhorn backward of several builds against the fp64 definition (autograd), ViT-B/16 head
geometry.  Dev tool, GPU only.   python tools/sinkhorn_accuracy.py base,product"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
dev = torch.device("cuda:0")
for (B, N, H, std) in [(4, 197, 12, 1.0), (4, 197, 12, 0.3), (2, 49, 12, 1.0), (2, 256, 4, 2.0)]:
    g = torch.Generator(device=dev).manual_seed(N)
    qkv = (torch.randn(B * N, 3 * H * 64, device=dev, generator=g) * std).bfloat16()
    do = (torch.randn(B * N, H * 64, device=dev, generator=g) * 0.5).bfloat16()
    qr = qkv.double().requires_grad_(True)
    q, k, v = qr.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    P = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1)
    for _ in range(3):
        P = P / P.sum(dim=-1, keepdim=True)
        P = P / P.sum(dim=-2, keepdim=True)
    P = P / P.sum(dim=-1, keepdim=True)
    o = (P @ v).permute(0, 2, 1, 3).reshape(B * N, H * 64)
    o.backward(do.double())
    ref = qr.grad.reshape(B * N, 3, H * 64)
    for l in libs:
        _devlib.use_library(l)
        out, lse, scal = K.attn_sinkhorn_fwd(qkv, B, N, H, 64, 0.125)
        d = K.attn_sinkhorn_bwd(qkv, do, lse, scal, B, N, H, 64, 0.125).double().reshape(B * N, 3, H * 64)
        parts = [((d[:, i] - ref[:, i]).norm() / ref[:, i].norm()).item() for i in range(3)]
        oe = ((out.double() - o.detach()).norm() / o.detach().norm()).item()
        print(f"B{B} N{N} H{H} std {std}: {l:8s} rel-L2 vs fp64: out {oe:.3e}  dq {parts[0]:.3e}  dk {parts[1]:.3e}  dv {parts[2]:.3e}", flush=True)
