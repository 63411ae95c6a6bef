#!/usr/bin/env python3
"""Per-shape timing of the MFMA GEMM kernels on the ViT-B/16 (batch 256) shapes.  GPU only; dev tool.

    python tools/gemm_bench.py                       # the product library
    python tools/gemm_bench.py product,exp1 [rounds]  # interleaved A/B of tools/_build/libnrv_hip_exp1.so vs the product
                                                      # in ONE process on ONE device (guide rule 24): median / min per build
Shapes: T = 50432 rows unless T=... in the environment; `only=nt|tn|misc` restricts the sections; `cold=1` evicts the caches
before every timed call (back-to-back repetitions keep ViT-S-sized operands in the 256 MB Infinity Cache, a training step does not)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import statistics
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *

dev = torch.device("cuda:0")
T = int(os.environ.get("T", 50432))
D = int(os.environ.get("D", 768))
only = os.environ.get("only", "")
libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
def rnd(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * 0.5).to(dt)

cold = os.environ.get("cold", "") == "1"       # cold=1: evict L2 / Infinity Cache before every timed call (in-step conditions:
_flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev) if cold else None     # the operands were not just read)
def timeit(fn, n=8):
    if cold:
        tot = 0.0
        for _ in range(n):
            _flush.add_(1)
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / n
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def ab(label, fn, flop=None, nbytes=None):
    """fn() is timed under every library in interleaved rounds."""
    res = {l: [] for l in libs}
    for l in libs:
        _devlib.use_library(l); fn(); fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for l in libs:
            _devlib.use_library(l)
            res[l].append(timeit(fn))
    parts = []
    for l in libs:
        med, mn = statistics.median(res[l]), min(res[l])
        rate = f"{flop/med/1e9:7.1f} TF" if flop else (f"{nbytes/med/1e6:6.0f} GB/s" if nbytes else "")
        parts.append(f"{l}: {med:7.4f} ms (min {mn:7.4f}) {rate}")
    print(f"{label:34s} " + " | ".join(parts), flush=True)

M4 = 4 * D
if only in ("", "nt"):
    for name, M, N, Kd, epi, odt in [
        ("qkv   bias->bf16", T, 3 * D, D, EPI_BIAS, torch.bfloat16),
        ("oproj resid->f32", T, D, D, EPI_BIAS_RESIDUAL, torch.float32),
        ("fc1   gelu->bf16", T, M4, D, EPI_BIAS_GELU, torch.bfloat16),
        ("fc2   resid->f32", T, D, M4, EPI_BIAS_RESIDUAL, torch.float32),
        ("dO    none->bf16", T, D, D, EPI_NONE, torch.bfloat16),
        ("dXn1  none->bf16", T, D, 3 * D, EPI_NONE, torch.bfloat16),
        ("dU    dgelu->bf16", T, M4, D, EPI_DGELU, torch.bfloat16),
        ("dXn2  none->bf16", T, D, M4, EPI_NONE, torch.bfloat16),
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
        ab(f"NT {name} {M}x{N}x{Kd}",
           lambda: K.gemm_nt(A, B, epilogue=epi, bias=bias if epi in (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL) else None,
                             aux=aux, aux_out=aux_out, out=out), flop=2.0 * M * N * Kd)
        del A, B, aux, aux_out, out
if only in ("", "tn"):
    for name, M, N in [("dWo", D, D), ("dWqkv", 3 * D, D), ("dW2", D, M4), ("dW1", M4, D)]:
        A = rnd(T, M); B = rnd(T, N)
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
        ab(f"TN {name} {M}x{N}xT", lambda: K.gemm_tn(A, B, out=out), flop=2.0 * M * N * T)
if only in ("", "misc"):
    H_ = D // 64
    B_, N_ = 256, 197
    qkv = rnd(B_ * N_, 3 * H_ * 64)
    ab("attn_fwd", lambda: K.attn_fwd(qkv, B_, N_, H_, 64, 0.125), nbytes=2.0 * B_ * N_ * H_ * 64 * 4)
    o, lse = K.attn_fwd(qkv, B_, N_, H_, 64, 0.125); do = rnd(B_ * N_, H_ * 64)
    ab("attn_bwd", lambda: K.attn_bwd(qkv, o, do, lse, B_, N_, H_, 64, 0.125), nbytes=2.0 * B_ * N_ * H_ * 64 * 12)
    x = rnd(T, D, dt=torch.float32); g = rnd(D, dt=torch.float32)
    ab("ln_fwd", lambda: K.layernorm_fwd(x, g, g, 1e-6), nbytes=T * D * 6.0)
    y, mu, rs = K.layernorm_fwd(x, g, g, 1e-6); dy = rnd(T, D)
    ab("ln_bwd", lambda: K.layernorm_bwd(dy, x, g, mu, rs, dres=x, want_f32=True, want_bf16=True), nbytes=T * D * 16.0)
    X = rnd(T, M4)
    ab("colsum [T,4D]", lambda: K.colsum(X), nbytes=X.numel() * 2.0)
