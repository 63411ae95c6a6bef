#!/usr/bin/env python3
"""Streaming attention kernels (N > 256 or head dim != 64) of several builds, interleaved.  Dev tool, GPU only.
    python tools/attn_gen_bench.py base,product"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
dev = torch.device("cuda:0")
def timeit(fn, n=6):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, B, N, H, dh in [("vit_b_16 @ 384 px", 32, 577, 12, 64), ("vit_h_14", 32, 257, 16, 80), ("simplevit dh 32", 64, 196, 24, 32),
                          ("dh 128, N 1024", 8, 1024, 8, 128)]:
    qkv = (torch.randn(B * N, 3 * H * dh, device=dev) * 0.5).bfloat16()
    do = (torch.randn(B * N, H * dh, device=dev) * 0.5).bfloat16()
    res = {l: ([], []) for l in libs}
    outs = {}
    for l in libs:
        _devlib.use_library(l)
        o, aux = K.attn_fwd(qkv, B, N, H, dh, dh ** -0.5)
        outs[l] = (o.float().clone(), K.attn_bwd(qkv, o, do, aux, B, N, H, dh, dh ** -0.5).float().clone())
    for l in libs[1:]:
        assert torch.equal(outs[l][0], outs[libs[0]][0]) and torch.equal(outs[l][1], outs[libs[0]][1]), f"{l} differs from {libs[0]}"
    for _ in range(4):
        for l in libs:
            _devlib.use_library(l)
            res[l][0].append(timeit(lambda: K.attn_fwd(qkv, B, N, H, dh, dh ** -0.5)))
            o, aux = K.attn_fwd(qkv, B, N, H, dh, dh ** -0.5)
            res[l][1].append(timeit(lambda: K.attn_bwd(qkv, o, do, aux, B, N, H, dh, dh ** -0.5)))
    for l in libs:
        print(f"{name:20s} B{B} N{N} H{H} dh{dh}: {l:8s} fwd {statistics.median(res[l][0]):7.3f} ms   bwd {statistics.median(res[l][1]):7.3f} ms", flush=True)
