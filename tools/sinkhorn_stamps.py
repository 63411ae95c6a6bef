#!/usr/bin/env python3
"""Phase times of the Sinkhorn backward (thread 0 of every workgroup, s_memtime) from an instrumented developer build
(`python tools/build_dev.py stamps --instrument`).  GPU only; dev tool.   python tools/sinkhorn_stamps.py stamps"""
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
import numpy as np
lib.nrv_dev_read_stamps_sinkhorn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nrv_dev_stamps_enable_sinkhorn() == 0
for _ in range(3):
    rc = lib.nrv_attn_sinkhorn_bwd(qkv.data_ptr(), do.data_ptr(), lse.data_ptr(), scal.data_ptr(), dqkv.data_ptr(),
                                   B, N, H, 64, 0.125, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
torch.cuda.synchronize()
buf = np.zeros(B * H * 16, dtype=np.uint64)
assert lib.nrv_dev_read_stamps_sinkhorn(buf.ctypes.data, buf.size) == 0
st = torch.from_numpy(buf.astype(np.int64)).view(B * H, 16)
st = st[st[:, 10] > 0]          # persistent kernel: one row per workgroup (its last head)
names = ["requests + P0", "wait + vectors + dV phase (P7 -> chunk -> MFMA)", "G init", "t=3 row+col", "t=2 row+col", "t=1 row+col", "t=0 row",
         "softmax backward", "dQ", "dK phase + next head's requests"]
tot = (st[:, 10] - st[:, 0]).double()
print(f"N={N}: {st.shape[0]} workgroups, head total median {tot.median().item():.0f} cycles")
for i, n in enumerate(names):
    d = (st[:, i + 1] - st[:, i]).double()
    print(f"  {n:48s} {d.median().item():8.0f} cycles  {100 * d.median().item() / tot.median().item():5.1f} %")
