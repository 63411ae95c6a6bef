#!/usr/bin/env python3
"""Stand-alone SinkhornAttention(scores) forward / backward of several builds on the shapes of the composed robust path.  Dev tool, GPU only.
    python tools/sinknorm_bench.py base,product"""
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
for name, G, N in [("vit_b_16 @ 384 px (32 x 12 heads)", 384, 577), ("vit_h_14 (32 x 16 heads)", 512, 257), ("module, 197 tokens", 768, 197), ("N 1000", 64, 1000)]:
    S = torch.randn(G, N, N, device=dev) * 1.5
    dP = torch.randn(G, N, N, device=dev)
    res = {l: ([], []) for l in libs}
    outs = {}
    for l in libs:
        _devlib.use_library(l)
        P, lse, a, b = K.sinkhorn_fwd(S, iters=3)
        outs[l] = (P.clone(), K.sinkhorn_bwd(S, dP, lse, a, b, iters=3).clone())
    for l in libs[1:]:
        assert torch.equal(outs[l][0], outs[libs[0]][0]) and torch.equal(outs[l][1], outs[libs[0]][1]), f"{l} differs"
    for _ in range(4):
        for l in libs:
            _devlib.use_library(l)
            res[l][0].append(timeit(lambda: K.sinkhorn_fwd(S, iters=3)))
            P, lse, a, b = K.sinkhorn_fwd(S, iters=3)
            res[l][1].append(timeit(lambda: K.sinkhorn_bwd(S, dP, lse, a, b, iters=3)))
    mb = G * N * N * 4 / 1e6
    for l in libs:
        f, bw = statistics.median(res[l][0]), statistics.median(res[l][1])
        print(f"{name:36s} {l:8s} fwd {f:7.3f} ms ({5 * mb / f / 1e3:4.2f} TB/s over 5 matrix passes)   bwd {bw:7.3f} ms ({12 * mb / bw / 1e3:4.2f} TB/s over 12 matrix passes)", flush=True)
