#!/usr/bin/env python3
"""NT GEMM of a developer build against the product on the test shapes (all epilogues): bit-equality (same summation order is
expected of schedule-only variants).  GPU only; dev tool.   python tools/nt_check.py LIB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
name = sys.argv[1]
def rnd(*s, dt=torch.bfloat16, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*s, generator=g, device=dev) * 0.5).to(dt)
bad = 0
for (M, N, Kd) in [(32, 192, 192), (300, 576, 192), (1000, 768, 768), (513, 384, 1536), (2048, 2304, 768), (1024, 3072, 768),
                   (512, 768, 3072), (50432, 768, 768), (25216, 1024, 4096), (12544, 768, 768), (4096, 4096, 4096), (257, 264, 256), (321, 72, 320),
                   (50432, 3072, 768), (50000, 1000, 704), (50432, 384, 384), (9000, 2048, 192)]:
    A = rnd(M, Kd, seed=1); B = rnd(N, Kd, seed=2); bias = rnd(N, dt=torch.float32, seed=3)
    res = rnd(M, N, dt=torch.float32, seed=4); du = rnd(M, N, seed=5)
    for epi, odt, kw in [(EPI_NONE, torch.bfloat16, {}), (EPI_NONE, torch.float32, {}), (EPI_BIAS, torch.bfloat16, dict(bias=bias)),
                         (EPI_BIAS_GELU, torch.bfloat16, dict(bias=bias)), (EPI_BIAS_RESIDUAL, torch.float32, dict(bias=bias, aux=res)),
                         (EPI_DGELU, torch.bfloat16, dict(aux=du))]:
        outs = []
        for lib in ("product", name):
            _devlib.use_library(lib)
            for rep in range(3):
                if epi == EPI_BIAS_GELU:              # with the gelu' stream: compare both outputs
                    u = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                    o = K.gemm_nt(A, B, epilogue=epi, out_dtype=odt, aux_out=u, **kw)
                    outs.append(torch.cat([o, u]).clone())
                    if rep == 0: outs.append(torch.cat([K.gemm_nt(A, B, epilogue=epi, out_dtype=odt, **kw), u]).clone())
                else:
                    outs.append(K.gemm_nt(A, B, epilogue=epi, out_dtype=odt, **kw).clone())
        ok = all(torch.equal(outs[0], o) for o in outs[1:])
        if not ok:
            bad += 1
            print("MISMATCH", (M, N, Kd), epi, odt, max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:]))
print("nt_check", name, "mismatches:", bad)
