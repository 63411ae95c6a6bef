#!/usr/bin/env python3
"""Target of a rocprofv3 --pmc pass: the dU launch ([50432 x 3072 x 768]) with the 8-bit / bf16 gelu' operand and without an epilogue, six launches
each (three kernel names in the counter CSV: gemm_nt8_kernel<..., 6 | 4 | 0, false, ...>).  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
T = 50432
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).bfloat16()
dY = rnd(T, 768); W = rnd(3072, 768)
out = torch.empty(T, 3072, dtype=torch.bfloat16, device=dev)
q = torch.randint(0, 255, (T, 3072), dtype=torch.uint8, device=dev); u = rnd(T, 3072)
for _ in range(6):
    K.gemm_nt(dY, W, epilogue=EPI_DGELU_Q8, aux=q, out=out)
    K.gemm_nt(dY, W, epilogue=EPI_DGELU, aux=u, out=out)
    K.gemm_nt(dY, W, epilogue=EPI_NONE, out=out)
torch.cuda.synchronize()
