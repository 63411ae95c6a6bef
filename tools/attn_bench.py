#!/usr/bin/env python3
"""Attention kernels only (ViT-B/16, batch 256): for rocprofv3 --kernel-trace --stats.  Dev tool, GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_robust_vit_amd import kernels as K
dev = torch.device("cuda:0")
B, N, H = int(os.environ.get("B", 256)), int(os.environ.get("N", 197)), int(os.environ.get("H", 12))
qkv = (torch.randn(B * N, 3 * H * 64, device=dev) * 0.5).bfloat16()
do = (torch.randn(B * N, H * 64, device=dev) * 0.5).bfloat16()
for _ in range(20):
    o, lse = K.attn_fwd(qkv, B, N, H, 64, 0.125)
    K.attn_bwd(qkv, o, do, lse, B, N, H, 64, 0.125)
torch.cuda.synchronize()
