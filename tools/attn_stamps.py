#!/usr/bin/env python3
"""Debug: per-phase s_memtime stamps of the persistent attention forward (NRV_ATTN_DBG=3).  Dev tool, GPU only."""
import os, sys, ctypes
os.environ["NRV_ATTN_DBG"] = str(4 | int(os.environ.get("DBG", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from noise_robust_vit_amd import kernels as K, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
lib.nrv_debug_read_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
B, N, H = 256, 197, 12
qkv = (torch.randn(B * N, 3 * H * 64, device=dev) * 0.5).bfloat16()
for _ in range(3): K.attn_fwd(qkv, B, N, H, 64, 0.125)
torch.cuda.synchronize()
buf = np.zeros(256 * 2 * 16 * 8, dtype=np.uint64)
assert lib.nrv_debug_read_attn_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(256, 2, 16, 8).astype(np.float64)
names = ["top->vmcnt0", "vmcnt0->barrier", "barrier->flush+issue", "S phase", "softmax", "PV", "end->next top"]
for w in range(2):
    t = s[:, w, :12, :]
    d = np.diff(t[:, :, :7], axis=2)            # [wg, it, 6]
    nxt = t[:, 1:, 0] - t[:, :-1, 6]
    print(f"wave {w*4}: cycles per head (median over WGs and heads 1..11)")
    for k in range(6):
        print(f"   {names[k]:24s} {np.median(d[:, 1:, k]):9.0f}")
    print(f"   {names[6]:24s} {np.median(nxt):9.0f}")
    print(f"   whole iteration          {np.median(t[:, 2:, 0] - t[:, 1:-1, 0]):9.0f}")
