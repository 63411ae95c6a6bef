#!/usr/bin/env python3
"""The batched strided GEMMs of the composed robust attention, several builds interleaved.  Dev tool, GPU only.
    python tools/bgemm_bench.py base,product"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, B, N, H, dh in [("vit_b_16 @ 384 px", 32, 577, 12, 64), ("vit_h_14", 32, 257, 16, 80)]:
    W = 3 * H * dh
    qkv = (torch.randn(B * N, W, device=dev) * 0.5).bfloat16()
    do = (torch.randn(B * N, H * dh, device=dev) * 0.5).bfloat16()
    rs, cs, bs, hs = W, 1, N * W, dh
    ors, ocs, obs, ohs = H * dh, 1, N * H * dh, dh
    mat, matT = (N, 1, H * N * N, N * N), (1, N, H * N * N, N * N)
    S = torch.empty(B, H, N, N, device=dev); P = torch.rand(B, H, N, N, device=dev)
    out = torch.empty(B * N, H * dh, dtype=torch.bfloat16, device=dev); dqkv = torch.empty_like(qkv)
    cases = {
        "S = q k^T   ": lambda: K.bgemm((qkv, 0), (rs, cs, bs, hs), (qkv, H * dh), (cs, rs, bs, hs), (S, 0), mat, B, H, N, N, dh, 0.125),
        "O = P v     ": lambda: K.bgemm((P, 0), mat, (qkv, 2 * H * dh), (rs, cs, bs, hs), (out, 0), (ors, ocs, obs, ohs), B, H, N, dh, N, 1.0),
        "dV = P^T dO ": lambda: K.bgemm((P, 0), matT, (do, 0), (ors, ocs, obs, ohs), (dqkv, 2 * H * dh), (rs, cs, bs, hs), B, H, N, dh, N, 1.0),
        "dP = dO v^T ": lambda: K.bgemm((do, 0), (ors, ocs, obs, ohs), (qkv, 2 * H * dh), (cs, rs, bs, hs), (S, 0), mat, B, H, N, N, dh, 1.0),
    }
    for cname, fn in cases.items():
        res = {l: [] for l in libs}
        for l in libs:
            _devlib.use_library(l); fn()
        for _ in range(4):
            for l in libs:
                _devlib.use_library(l); res[l].append(timeit(fn))
        print(f"{name:18s} {cname} " + "   ".join(f"{l} {statistics.median(res[l]):6.3f} ms" for l in libs), flush=True)
