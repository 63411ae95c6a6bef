#!/usr/bin/env python3
"""Sinkhorn attention forward / backward only (ViT-B/16, batch 256) for several builds, interleaved.  Dev tool, GPU only."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
B, N, H = int(os.environ.get("B", 256)), int(os.environ.get("N", 197)), int(os.environ.get("H", 12))
dev = torch.device("cuda:0")
qkv = (torch.randn(B * N, 3 * H * 64, device=dev) * 0.5).bfloat16()
do = (torch.randn(B * N, H * 64, device=dev) * 0.5).bfloat16()
def timeit(fn, n=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
res = {l: ([], []) for l in libs}
for l in libs:
    _devlib.use_library(l)
    o, lse, scal = K.attn_sinkhorn_fwd(qkv, B, N, H, 64, 0.125); K.attn_sinkhorn_bwd(qkv, do, lse, scal, B, N, H, 64, 0.125)
torch.cuda.synchronize()
for _ in range(4):
    for l in libs:
        _devlib.use_library(l)
        res[l][0].append(timeit(lambda: K.attn_sinkhorn_fwd(qkv, B, N, H, 64, 0.125)))
        o, lse, scal = K.attn_sinkhorn_fwd(qkv, B, N, H, 64, 0.125)
        res[l][1].append(timeit(lambda: K.attn_sinkhorn_bwd(qkv, do, lse, scal, B, N, H, 64, 0.125)))
for l in libs:
    print(f"{l:12s} sinkhorn fwd {statistics.median(res[l][0]):7.3f} ms   bwd {statistics.median(res[l][1]):7.3f} ms")
