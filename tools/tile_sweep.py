#!/usr/bin/env python3
"""NT GEMM tile-height sweep: the forced-tile developer builds (tools/build_dev.py tNNN --instrument -DNRV_DEV_NO_STAMPS -DNRV_FORCE_NT_TILE=NNN) timed in
interleaved rounds on the shapes of every BASELINE config; prints the fastest height per shape next to what the product's
cost model (`nt_tile_choice`) picks.  GPU only; dev tool."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *

dev = torch.device("cuda:0")
libs = os.environ.get("libs", "t128,t192,t256,t320").split(",") + ["product"]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
def rnd(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * 0.5).to(dt)
def timeit(fn, n=6):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def shapes_for(T, D, M4):
    return [("qkv bias", T, 3 * D, D, EPI_BIAS, torch.bfloat16), ("oproj resid", T, D, D, EPI_BIAS_RESIDUAL, torch.float32),
            ("fc1 gelu", T, M4, D, EPI_BIAS_GELU, torch.bfloat16), ("fc2 resid", T, D, M4, EPI_BIAS_RESIDUAL, torch.float32),
            ("dO none", T, D, D, EPI_NONE, torch.bfloat16), ("dXn1 none", T, D, 3 * D, EPI_NONE, torch.bfloat16),
            ("dU dgelu", T, M4, D, EPI_DGELU, torch.bfloat16), ("dXn2 none", T, D, M4, EPI_NONE, torch.bfloat16)]

sets = {"vit_b (T 50432)": shapes_for(50432, 768, 3072), "mae_b encoder (T 12544)": shapes_for(12544, 768, 3072),
        "mae decoder (T 50176, D 512)": shapes_for(50176, 512, 2048), "vit_s (T 50432)": shapes_for(50432, 384, 1536),
        "vit_l (T 25216)": shapes_for(25216, 1024, 4096)}
which = os.environ.get("sets", "")
for title, shapes in sets.items():
    if which and not any(w in title for w in which.split(",")):
        continue
    print(f"== {title}")
    for name, M, N, Kd, epi, odt in shapes:
        A = rnd(M, Kd); B = rnd(N, Kd); bias = rnd(N, dt=torch.float32)
        aux = aux_out = None
        if epi == EPI_BIAS_RESIDUAL: aux = rnd(M, N, dt=torch.float32)
        if epi == EPI_DGELU: aux = rnd(M, N)
        if epi == EPI_BIAS_GELU: aux_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out = torch.empty(M, N, dtype=odt, device=dev)
        fn = lambda: K.gemm_nt(A, B, epilogue=epi, bias=bias if epi in (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL) else None,
                               aux=aux, aux_out=aux_out, out=out)
        res = {l: [] for l in libs}
        outs = {}
        for l in libs:
            _devlib.use_library(l); fn(); fn()
            outs[l] = out.float().clone()
        torch.cuda.synchronize()
        bad = [l for l in libs if not torch.equal(outs[l], outs["product"])]       # same K order in every tile shape: bit-equal
        if bad: print(f"  !! {name}: results of {bad} differ from the product build "
                      f"(max {max(float((outs[l] - outs['product']).abs().max()) for l in bad):.3e})")
        del outs
        for _ in range(rounds):
            for l in libs:
                _devlib.use_library(l); res[l].append(timeit(fn))
        med = {l: statistics.median(v) for l, v in res.items()}
        best = min(libs[:-1], key=lambda l: med[l])
        print(f"  {name:12s} {M}x{N}x{Kd}: " + "  ".join(f"{l} {med[l]*1e3:7.1f}us" for l in libs) +
              f"  | best {best}, product/best = {med['product']/med[best]:.3f}", flush=True)
        del A, B, aux, aux_out, out
