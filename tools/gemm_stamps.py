#!/usr/bin/env python3
"""Per-workgroup phase timing of the NT GEMM.  Dev tool, GPU only: needs the instrumented developer build
(`python tools/build_dev.py stamps -DNRV_DEV_STAMPS`, built in the dev container; the product library has no stamps)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _devlib
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd._lib import *
dev = torch.device("cuda:0")
lib = _devlib.use_library("stamps")
lib.nrv_dev_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
T = 50432
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
    import math
    tn = -(-N // 256)
    cands = [(256, 320.0), (320, 384.0 * 1.01), (192, 256.0), (128, 192.0)]      # nt_tile_choice (nrv_gemm.hip)
    best = None
    for h, pt in cands:
        c = math.ceil(-(-M // h) * tn / 256) * pt
        if best is None or c < best[1] * 0.999: best = (h, c)
    nwg = -(-M // best[0]) * tn
    buf = np.zeros(nwg * 5, dtype=np.uint64)
    assert lib.nrv_dev_read_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.reshape(nwg, 5)
    t = s[:, :4].astype(np.float64) * 0.01     # 100 MHz -> us
    base = t[:, 0].min()
    pro, loop, epi_t = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"{name}: tile {best[0]}x256, wgs {nwg} kernel span {t[:,3].max()-base:.1f} us | prologue {pro.mean():.2f} (max {pro.max():.1f}) "
          f"loop {loop.mean():.2f} (min {loop.min():.1f} max {loop.max():.1f}) epilogue {epi_t.mean():.2f} (max {epi_t.max():.1f}) us")
    start = np.sort(t[:, 0] - base)
    print("   start times percentiles us:", np.percentile(start, [0, 25, 50, 75, 100]).round(1))
    wbuf = np.zeros((1 << 19) + nwg * 8 * 3, dtype=np.uint64)
    assert lib.nrv_dev_read_stamps(wbuf.ctypes.data, wbuf.size) == 0
    w = wbuf[1 << 19:].reshape(nwg, 8, 3).astype(np.float64)
    nk = -(-Kd // 64)
    if nk > 1:
        vm, bar, tot = w[..., 0].mean(), w[..., 1].mean(), w[..., 2].mean()
        print(f"   K-tiles 1..{nk-1}, per wave (shader cycles): total {tot:.0f} = {tot/(nk-1):.0f} per K-tile | parked on vmcnt(0) {vm:.0f} "
              f"({100*vm/tot:.1f} %) | on the barrier {bar:.0f} ({100*bar/tot:.1f} %) | waves 0-3 vm {w[:, :4, 0].mean():.0f} bar {w[:, :4, 1].mean():.0f}, waves 4-7 vm {w[:, 4:, 0].mean():.0f} bar {w[:, 4:, 1].mean():.0f}")
for name, M, N, Kd, epi, odt in [("dO none", T, 768, 768, 0, torch.bfloat16), ("qkv bias", T, 2304, 768, 1, torch.bfloat16),
                                  ("oproj resid f32", T, 768, 768, 3, torch.float32), ("fc1 gelu", T, 3072, 768, 2, torch.bfloat16),
                                  ("dXn2 none K3072", T, 768, 3072, 0, torch.bfloat16), ("sq8192", 8192, 8192, 8192, 0, torch.bfloat16)]:
    run(M, N, Kd, epi, odt, name)
