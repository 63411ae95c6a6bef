#!/usr/bin/env python3
"""Where the fc1 (GELU) epilogue's time goes: same GEMM with (a) both outputs, (b) no gelu' output, (c) bias only.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
if len(sys.argv) > 1:                      # a developer build of the library (tools/build_dev.py NAME)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _devlib
    _devlib.use_library(sys.argv[1])
T = 50432
def rnd(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * 0.5).to(dt)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
A = rnd(T, 768); B = rnd(3072, 768); bias = rnd(3072, dt=torch.float32)
out = torch.empty(T, 3072, dtype=torch.bfloat16, device=dev); aux_out = torch.empty_like(out)
for rep in range(2):
    a = timeit(lambda: K.gemm_nt(A, B, epilogue=EPI_BIAS_GELU, bias=bias, aux_out=aux_out, out=out))
    q = torch.empty(T, 3072, dtype=torch.uint8, device=dev)
    a8 = timeit(lambda: K.gemm_nt(A, B, epilogue=EPI_BIAS_GELU_Q8, bias=bias, aux_out=q, out=out))
    b = timeit(lambda: K.gemm_nt(A, B, epilogue=EPI_BIAS_GELU, bias=bias, out=out))
    c = timeit(lambda: K.gemm_nt(A, B, epilogue=EPI_BIAS, bias=bias, out=out))
    d = timeit(lambda: K.gemm_nt(A, B, epilogue=EPI_NONE, out=out))
    print(f"fc1 shape: gelu + gelu' {a:.3f} ms | gelu + 8-bit gelu' {a8:.3f} | gelu only {b:.3f} | bias only {c:.3f} | none {d:.3f}")
dY = rnd(T, 768); W2t = rnd(3072, 768)
for rep in range(2):
    a = timeit(lambda: K.gemm_nt(dY, W2t, epilogue=EPI_DGELU, aux=aux_out, out=out))
    a8 = timeit(lambda: K.gemm_nt(dY, W2t, epilogue=EPI_DGELU_Q8, aux=q, out=out))
    d = timeit(lambda: K.gemm_nt(dY, W2t, epilogue=EPI_NONE, out=out))
    print(f"dU shape: dgelu (bf16 stream) {a:.3f} ms | dgelu (8-bit stream) {a8:.3f} | none {d:.3f}")
