#!/usr/bin/env python3
"""Per-shape timing of the MFMA GEMM kernels on the ViT-B/16 (batch 256) shapes.  GPU only; dev tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *

dev = torch.device("cuda:0")
T = int(os.environ.get("T", 50432))
def rnd(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * 0.5).to(dt)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

rows = []
for name, M, N, Kd, epi, odt in [
    ("qkv   bias->bf16", T, 2304, 768, EPI_BIAS, torch.bfloat16),
    ("oproj resid->f32", T, 768, 768, EPI_BIAS_RESIDUAL, torch.float32),
    ("fc1   gelu->bf16", T, 3072, 768, EPI_BIAS_GELU, torch.bfloat16),
    ("fc2   resid->f32", T, 768, 3072, EPI_BIAS_RESIDUAL, torch.float32),
    ("dO    none->bf16", T, 768, 768, EPI_NONE, torch.bfloat16),
    ("dXn1  none->bf16", T, 768, 2304, EPI_NONE, torch.bfloat16),
    ("dU    dgelu->bf16", T, 3072, 768, EPI_DGELU, torch.bfloat16),
    ("dXn2  none->bf16", T, 768, 3072, EPI_NONE, torch.bfloat16),
    ("sq4096 none->bf16", 4096, 4096, 4096, EPI_NONE, torch.bfloat16),
    ("sq8192 none->bf16", 8192, 8192, 8192, EPI_NONE, torch.bfloat16),
]:
    A = rnd(M, Kd); B = rnd(N, Kd)
    bias = rnd(N, dt=torch.float32)
    aux = None; aux_out = None
    if epi == EPI_BIAS_RESIDUAL: aux = rnd(M, N, dt=torch.float32)
    if epi == EPI_DGELU: aux = rnd(M, N)
    if epi == EPI_BIAS_GELU: aux_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=odt, device=dev)
    ms = timeit(lambda: K.gemm_nt(A, B, epilogue=epi, bias=bias if epi in (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL) else None,
                                  aux=aux, aux_out=aux_out, out=out))
    print(f"NT {name:20s} M={M:6d} N={N:5d} K={Kd:5d}  {ms:8.3f} ms  {2*M*N*Kd/ms/1e9:8.1f} TFLOP/s")
    del A, B, aux, aux_out, out
for name, M, N in [("dWo", 768, 768), ("dWqkv", 2304, 768), ("dW2", 768, 3072), ("dW1", 3072, 768)]:
    A = rnd(T, M); B = rnd(T, N)
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ms = timeit(lambda: K.gemm_tn(A, B, out=out))
    print(f"TN {name:20s} M={M:6d} N={N:5d} T={T:5d}  {ms:8.3f} ms  {2*M*N*T/ms/1e9:8.1f} TFLOP/s")
X = rnd(T, 3072)
ms = timeit(lambda: K.colsum(X)); print(f"colsum [T,3072] {ms:.3f} ms {X.numel()*2/ms/1e6:.0f} GB/s")
X = rnd(T, 768)
ms = timeit(lambda: K.colsum(X)); print(f"colsum [T,768] {ms:.3f} ms {X.numel()*2/ms/1e6:.0f} GB/s")
B_, N_, H_ = 256, 197, 12
qkv = rnd(B_*N_, 3*H_*64)
ms = timeit(lambda: K.attn_fwd(qkv, B_, N_, H_, 64, 0.125)); print(f"attn_fwd {ms:.3f} ms {4*B_*H_*N_*N_*64/ms/1e9:.1f} TFLOP/s")
o, lse = K.attn_fwd(qkv, B_, N_, H_, 64, 0.125); do = rnd(B_*N_, H_*64)
ms = timeit(lambda: K.attn_bwd(qkv, o, do, lse, B_, N_, H_, 64, 0.125)); print(f"attn_bwd {ms:.3f} ms {10*B_*H_*N_*N_*64/ms/1e9:.1f} TFLOP/s")
x = rnd(T, 768, dt=torch.float32); g = rnd(768, dt=torch.float32)
ms = timeit(lambda: K.layernorm_fwd(x, g, g, 1e-6)); print(f"ln_fwd {ms:.3f} ms {T*768*6/ms/1e6:.0f} GB/s")
y, mu, rs = K.layernorm_fwd(x, g, g, 1e-6); dy = rnd(T, 768)
ms = timeit(lambda: K.layernorm_bwd(dy, x, g, mu, rs, dres=x, want_f32=True, want_bf16=True)); print(f"ln_bwd {ms:.3f} ms {T*768*16/ms/1e6:.0f} GB/s")
