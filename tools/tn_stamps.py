#!/usr/bin/env python3
"""Wave accounting of the phased TN (weight-gradient) kernel: shader cycles per K-step and phase spent in section 0 (fragment
reads + DMA issue), 2 (counted vmcnt wait + first barrier + fragment-read latency), 1 (the 16 MFMAs of the quadrant issued), 3 (bias-gradient MFMAs + closing barrier), waves
0-3 and 4-7.  Dev tool, GPU only: needs the instrumented build (`python tools/build_dev.py stamps --instrument`)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _devlib
from noise_robust_vit_amd import kernels as K
dev = torch.device("cuda:0")
lib = _devlib.use_library(sys.argv[1] if len(sys.argv) > 1 else "stamps")
lib.nrv_dev_read_stamps_gemm.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nrv_dev_stamps_enable_gemm() == 0
T = int(os.environ.get("T", 50432))
for name, M, N, bias in [("dWqkv", 2304, 768, True), ("dWo", 768, 768, True), ("dW1", 3072, 768, True), ("dW2", 768, 3072, True), ("dW1 no bias", 3072, 768, False)]:
    A = (torch.randn(T, M, device=dev) * .5).bfloat16(); B = (torch.randn(T, N, device=dev) * .5).bfloat16()
    f = lambda: K.gemm_tn(A, B, want_dbias=bias)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    tiles = -(-M // 256) * -(-N // 256)
    splits = max(1, min(256 // tiles, -(-T // 64)))
    nwg = tiles * splits
    nk = -(-T // 64) / splits
    buf = np.zeros((1 << 19) + nwg * 128, dtype=np.uint64)
    assert lib.nrv_dev_read_stamps_gemm(buf.ctypes.data, buf.size) == 0
    w = buf[1 << 19:].reshape(nwg, 8, 4, 4).astype(np.float64) / nk
    print(f"{name}: [{M} x {N}] over {T} tokens, {tiles} tiles x {splits} splits, {nk:.1f} K-steps each; launch {s.elapsed_time(e) * 1e3:.1f} us (instrumented); "
          f"{w.sum((-1, -2)).mean():.0f} cycles per K-step; sections [issue, 16 MFMAs, vmcnt+barrier+lgkm, bias+closing barrier] (executed in the order 0, 2, 1, 3):")
    for P in range(4):
        g0, g1 = w[:, :4, P].mean((0, 1)), w[:, 4:, P].mean((0, 1))
        print(f"      phase {P}: waves 0-3 {g0.round(0)} = {g0.sum():.0f} | waves 4-7 {g1.round(0)} = {g1.sum():.0f}")
    if os.environ.get("per_wave"):
        for P in range(4):
            print(f"      phase {P} per wave (sections 0, 1, 2, 3): " + " | ".join(f"w{v}: " + " ".join(f"{x:.0f}" for x in w[:, v, P].mean(0)) for v in range(8)))
