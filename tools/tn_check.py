#!/usr/bin/env python3
"""TN GEMM (weight gradient + fused bias gradient) of a developer build against the product build: same K order, so the
results must be bit-equal.  Covers 1, 2, 3 and many K-tiles per split, ragged token counts, the row-remapped A operand.
GPU only; dev tool.   python tools/tn_check.py LIB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
lib = sys.argv[1]
dev = torch.device("cuda:0")
bad = 0
for (T, M, N) in [(50432, 768, 768), (50432, 3072, 768), (3000, 576, 192), (130, 256, 256), (64, 256, 256), (65, 256, 256), (1, 256, 256),
                  (128, 256, 512), (129, 512, 256), (192, 256, 256), (777, 104, 72), (4096, 768, 3072), (12544, 1536, 512), (200, 2304, 768)]:
    g = torch.Generator(device=dev).manual_seed(T + M)
    A = (torch.randn(T, M, device=dev, generator=g) * 0.5).bfloat16()
    B = (torch.randn(T, N, device=dev, generator=g) * 0.5).bfloat16()
    outs = {}
    for l in ("product", lib):
        _devlib.use_library(l)
        c, db = K.gemm_tn(A, B, want_dbias=True)
        outs[l] = (c.clone(), db.clone())
    torch.cuda.synchronize()
    ok = torch.equal(outs["product"][0], outs[lib][0]) and torch.equal(outs["product"][1], outs[lib][1])
    ref = A.float().t() @ B.float()
    rel = ((outs[lib][0] - ref).norm() / ref.norm()).item()
    print(f"T {T:6d} M {M:5d} N {N:5d}: bit-equal to product {ok}   rel-L2 vs fp32 torch {rel:.2e}", flush=True)
    bad += not ok
# row-remapped A (class-token slot)
Bt, G, D, F = 3, 196, 128, 64
dY = (torch.randn(Bt * (G + 1), D, device=dev) * 0.5).bfloat16(); P = (torch.randn(Bt * G, F, device=dev) * 0.5).bfloat16()
r = {}
for l in ("product", lib):
    _devlib.use_library(l); r[l] = K.gemm_tn(dY, P, a_group=G, a_group_stride=G + 1, a_row_offset=1, T=Bt * G).clone()
ok = torch.equal(r["product"], r[lib]); print("row remap bit-equal", ok); bad += not ok
sys.exit(1 if bad else 0)
