#!/usr/bin/env python3
"""Per-workgroup phase timing of the NT GEMM.  Dev tool, GPU only: needs the instrumented developer build
(`python tools/build_dev.py stamps --instrument`, built in the dev container; the product library has no stamps).

Workgroup stamps: prologue (launch -> first K-step ready), K loop, epilogue.  Wave accounting of the phased K loop
(gemm_nt8_kernel): shader cycles per K-step spent in section 0 (fragment-read + DMA issue), 1 (counted vmcnt wait),
2 (first barrier + fragment-read latency), 3 (MFMA section + closing barrier), for waves 0-3 and 4-7."""
import os, sys, ctypes, math
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _devlib
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
lib = _devlib.use_library(sys.argv[1] if len(sys.argv) > 1 else "stamps")
lib.nrv_dev_read_stamps_gemm.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nrv_dev_stamps_enable_gemm() == 0
T = int(os.environ.get("T", 50432))

def tile_rows(M, N):
    tn = -(-N // 256)
    cands = [(256, 320.0), (320, 384.0 * 1.01), (192, 256.0), (128, 192.0)]      # nt_tile_choice (nrv_gemm.hip)
    best = None
    for h, pt in cands:
        c = math.ceil(-(-M // h) * tn / 256) * pt
        if best is None or c < best[1] * 0.999: best = (h, c)
    return best[0], -(-M // best[0]) * tn

def run(M, N, Kd, epi, odt, name):
    A = (torch.randn(M, Kd, device=dev) * .5).bfloat16(); B = (torch.randn(N, Kd, device=dev) * .5).bfloat16()
    bias = torch.randn(N, device=dev); aux = None; aux_out = None
    if epi == EPI_BIAS_RESIDUAL: aux = torch.randn(M, N, device=dev)
    if epi == EPI_DGELU: aux = torch.randn(M, N, device=dev).bfloat16()
    if epi == EPI_BIAS_GELU: aux_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=odt, device=dev)
    f = lambda: K.gemm_nt(A, B, epilogue=epi, bias=bias if epi in (1, 2, 3) else None, aux=aux, aux_out=aux_out, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    rows, nwg = tile_rows(M, N)
    buf = np.zeros((1 << 19) + nwg * 8 * 16, dtype=np.uint64)
    assert lib.nrv_dev_read_stamps_gemm(buf.ctypes.data, buf.size) == 0
    s = buf[:nwg * 8].reshape(nwg, 8)
    t = s[:, :4].astype(np.float64) * 0.01     # 100 MHz -> us
    base = t[:, 0].min()
    pro, loop, epi_t = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    nk = -(-Kd // 64)
    print(f"{name}: tile {rows}x256, wgs {nwg} kernel span {t[:,3].max()-base:.1f} us | prologue {pro.mean():.2f} (max {pro.max():.1f}) "
          f"loop {loop.mean():.2f} (min {loop.min():.1f} max {loop.max():.1f}; {loop.mean()/nk:.3f} per K-step) epilogue {epi_t.mean():.2f} (max {epi_t.max():.1f}) us")
    start = np.sort(t[:, 0] - base)
    print("   start times percentiles us:", np.percentile(start, [0, 25, 50, 75, 100]).round(1))
    w = buf[1 << 19:].reshape(nwg, 8, 4, 4).astype(np.float64)
    tot = w.sum((-1, -2)).mean()
    print(f"   K loop per wave {tot:.0f} shader cycles = {tot/nk:.0f} per K-step ({tot/loop.mean()/1e3:.2f} GHz); per K-step and phase, "
          f"sections [issue, vmcnt, barrier+lgkm, mfma+barrier]:")
    for P in range(4):
        g0, g1 = w[:, :4, P].mean((0, 1)) / nk, w[:, 4:, P].mean((0, 1)) / nk
        print(f"      phase {P}: waves 0-3 {g0.round(0)} = {g0.sum():.0f} | waves 4-7 {g1.round(0)} = {g1.sum():.0f}")

for name, M, N, Kd, epi, odt in [("dO none", T, 768, 768, 0, torch.bfloat16), ("qkv bias", T, 2304, 768, 1, torch.bfloat16),
                                  ("oproj resid f32", T, 768, 768, 3, torch.float32), ("fc1 gelu", T, 3072, 768, 2, torch.bfloat16),
                                  ("dU dgelu", T, 3072, 768, 4, torch.bfloat16),
                                  ("dXn2 none K3072", T, 768, 3072, 0, torch.bfloat16), ("sq8192", 8192, 8192, 8192, 0, torch.bfloat16)]:
    run(M, N, Kd, epi, odt, name)
