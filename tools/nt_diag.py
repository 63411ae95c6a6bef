#!/usr/bin/env python3
"""Where does a developer build's NT GEMM differ from another build?  python tools/nt_diag.py LIB_A LIB_B M N K   (fp32 output, no epilogue)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
la, lb = sys.argv[1], sys.argv[2]
M, N, Kd = (int(v) for v in sys.argv[3:6])
g = torch.Generator(device=dev).manual_seed(1)
A = (torch.randn(M, Kd, generator=g, device=dev) * 0.5).bfloat16(); B = (torch.randn(N, Kd, generator=g, device=dev) * 0.5).bfloat16()
outs = {}
for rep in range(3):
    for l in (la, lb):
        _devlib.use_library(l)
        outs[(l, rep)] = K.gemm_nt(A, B, out_dtype=torch.float32).clone()
ref = A.float() @ B.float().t()
for l in (la, lb):
    for rep in range(3):
        o = outs[(l, rep)]
        bad = (o - ref).abs() > 2e-2
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print(f"{l} rep {rep}: {int(bad.sum())} wrong of {o.numel()}; rows {rows[:6]}..{rows[-3:]} ({len(rows)}), cols {cols[:6]}..{cols[-3:]} ({len(cols)})")
        if bad.any():
            i, j = bad.nonzero()[0].tolist()
            print(f"    first wrong [{i},{j}] got {o[i, j].item():.6g} want {ref[i, j].item():.4f}; zero? {(o[bad] == 0).float().mean().item():.2f} of wrong are 0")
            ij = bad.nonzero()
            import collections
            blocks = collections.Counter((int(a) // 16, int(b) // 64) for a, b in ij.tolist())
            for (rb, cb), n in list(blocks.items())[:8]:
                sub = bad[rb * 16:(rb + 1) * 16, cb * 64:(cb + 1) * 64]
                rr = sub.any(1).nonzero().flatten().tolist(); cc = sub.any(0).nonzero().flatten().tolist()
                print(f"      slab rows {rb * 16}.. cols {cb * 64}..: {n} wrong; rows-in-slab {rr}; cols-in-block {cc[:20]}{'...' if len(cc) > 20 else ''}; "
                      f"got sample {[round(v, 4) for v in o[rb * 16 + rr[0], cb * 64:(cb + 1) * 64][sub[rr[0]]].tolist()[:6]]}")
