#!/usr/bin/env python3
"""Phase times of the Sinkhorn backward (wave 0 of every workgroup, s_memtime) from a -DNRV_SK_STAMPS developer build
(`python tools/build_dev.py skst -DNRV_SK_STAMPS`).  GPU only; dev tool.   python tools/sinkhorn_stamps.py skst"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K, _lib
B, N, H = 256, int(os.environ.get("N", 197)), 12
dev = torch.device("cuda:0")
qkv = (torch.randn(B * N, 3 * H * 64, device=dev) * 0.5).bfloat16()
do = (torch.randn(B * N, H * 64, device=dev) * 0.5).bfloat16()
_devlib.use_library(sys.argv[1])
lib = _lib.load()
o, lse, scal = K.attn_sinkhorn_fwd(qkv, B, N, H, 64, 0.125)
dqkv = torch.empty_like(qkv)
import ctypes
ws = torch.zeros(B * H * 128, dtype=torch.uint8, device=dev)
lib.nrv_dev_sinkhorn_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.nrv_dev_sinkhorn_stamp_buffer.restype = None
lib.nrv_dev_sinkhorn_stamp_buffer(ws.data_ptr())
for _ in range(3):
    rc = lib.nrv_attn_sinkhorn_bwd(qkv.data_ptr(), do.data_ptr(), lse.data_ptr(), scal.data_ptr(), dqkv.data_ptr(),
                                   B, N, H, 64, 0.125, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
torch.cuda.synchronize()
st = ws[:B * H * 128].view(torch.int64).view(B * H, 16).cpu()
names = ["images K/dO + vectors", "dV phase (P7 -> chunk -> MFMA)", "V image + G init", "t=3 row+col", "t=2 row+col", "t=1 row+col", "t=0 row",
         "softmax backward", "dQ", "Q image + dK phase"]
tot = (st[:, 10] - st[:, 0]).double()
print(f"N={N}: workgroup total median {tot.median().item():.0f} cycles")
for i, n in enumerate(names):
    d = (st[:, i + 1] - st[:, i]).double()
    print(f"  {n:34s} {d.median().item():8.0f} cycles  {100 * d.median().item() / tot.median().item():5.1f} %")
