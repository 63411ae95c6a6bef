#!/usr/bin/env python3
"""LayerNorm forward / backward of several builds at the BASELINE widths, interleaved.  Dev tool, GPU only.
    python tools/ln_bench.py product,lnA,...   (variants built with another LN_*_BLOCKS constant)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for T, D in [(50432, 384), (50432, 768), (25216, 1024), (12800, 768), (50176, 512)]:
    x = torch.randn(T, D, device=dev); w = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    dy = torch.randn(T, D, device=dev).bfloat16(); dres = torch.randn(T, D, device=dev)
    res = {l: ([], []) for l in libs}
    for l in libs:
        _devlib.use_library(l)
        xn, mean, rstd = K.layernorm_fwd(x, w, b, 1e-6)
        K.layernorm_bwd(dy, x, w, mean, rstd, dres=dres, want_f32=True, want_bf16=True)
    for _ in range(5):
        for l in libs:
            _devlib.use_library(l)
            res[l][0].append(timeit(lambda: K.layernorm_fwd(x, w, b, 1e-6)))
            res[l][1].append(timeit(lambda: K.layernorm_bwd(dy, x, w, mean, rstd, dres=dres, want_f32=True, want_bf16=True)))
    fb, bb = T * D * 6, T * D * 16
    for l in libs:
        f, bw = statistics.median(res[l][0]), statistics.median(res[l][1])
        print(f"[{T} x {D}] {l:8s} fwd {f:7.1f} us ({fb / f / 1e6:5.2f} TB/s)   bwd {bw:7.1f} us ({bb / bw / 1e6:5.2f} TB/s)", flush=True)
