#!/usr/bin/env python3
"""dU launch ([50432 x 3072 x 768], NRV_EPI_DGELU_Q8) under several developer builds in one process.  Dev tool.
    python tools/du_epilogue_split.py product,v1,v2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
T = 50432
libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
def rnd(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * 0.5).to(dt)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
dY = rnd(T, 768); W = rnd(3072, 768)
out = torch.empty(T, 3072, dtype=torch.bfloat16, device=dev)
q = torch.randint(0, 255, (T, 3072), dtype=torch.uint8, device=dev); u = rnd(T, 3072)
for rep in range(3):
    for l in libs:
        _devlib.use_library(l)
        a = timeit(lambda: K.gemm_nt(dY, W, epilogue=EPI_DGELU_Q8, aux=q, out=out))
        b = timeit(lambda: K.gemm_nt(dY, W, epilogue=EPI_DGELU, aux=u, out=out))
        d = timeit(lambda: K.gemm_nt(dY, W, epilogue=EPI_NONE, out=out))
        print(f"{l:8s} dU: 8-bit stream {a:.3f} ms | bf16 stream {b:.3f} | none {d:.3f}", flush=True)
