#!/usr/bin/env python3
"""Per-workgroup phase timing of the NT GEMM.  Dev tool, GPU only: needs the instrumented developer build
(`python tools/build_dev.py stamps --instrument`, built in the dev container; the product library has no stamps).

Tile stamps of the persistent kernel (thread 0 of every workgroup, along its tile sequence): K loop, wave-group re-alignment,
epilogue, hand-over to the next tile.  Wave accounting of the phased K loop
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
    cands = [(256, 320.0), (320, 320.0 * 1.3), (192, 256.0), (128, 192.0)]      # nt_tile_choice (nrv_gemm.hip)
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
    if epi == EPI_DGELU_Q8: aux = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev)
    if epi == EPI_BIAS_GELU: aux_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=odt, device=dev)
    f = lambda: K.gemm_nt(A, B, epilogue=epi, bias=bias if epi in (1, 2, 3) else None, aux=aux, aux_out=aux_out, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    rows, ntiles = tile_rows(M, N)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count & ~7
    nwg = min(ntiles, cus)                     # persistent kernel: one workgroup per CU walks tiles b, b + nwg, ...
    buf = np.zeros((1 << 20) + nwg * 32, dtype=np.uint64)
    assert lib.nrv_dev_read_stamps_gemm(buf.ctypes.data, buf.size) == 0
    ts = buf[1 << 20:].reshape(nwg, 32).astype(np.float64) * 0.01      # 100 MHz -> us; [0] start, [1 + 4 i .. 4 + 4 i] tile i, then "drained"
    base = ts[:, 0].min()
    nk = -(-Kd // 64)
    per_wg = [(ntiles - b + nwg - 1) // nwg for b in range(nwg)]
    end = np.array([ts[b, min(1 + 4 * per_wg[b], 31)] for b in range(nwg)])
    print(f"{name}: tile {rows}x256, {ntiles} tiles on {nwg} workgroups ({min(per_wg)}-{max(per_wg)} each), kernel span {end.max() - base:.1f} us; "
          f"first K-step ready {np.mean(ts[:, 1] - ts[:, 0]):.2f} us after start")
    for i in range(min(max(per_wg), 7)):
        sel = [b for b in range(nwg) if per_wg[b] > i]
        t0, t1, t2, t3 = (ts[sel, 1 + 4 * i + j] for j in range(4))
        nxt = np.array([ts[b, 5 + 4 * i] if per_wg[b] > i + 1 and 5 + 4 * i < 32 else np.nan for b in sel])
        print(f"   tile {i} ({len(sel)} workgroups): starts at {np.mean(t0 - base):7.1f} us (spread {np.ptp(t0):5.1f}) | K loop {np.mean(t1 - t0):6.2f} ({np.mean(t1 - t0) / nk:.3f} per K-step) "
              f"| re-align {np.mean(t2 - t1):5.2f} | epilogue {np.mean(t3 - t2):5.2f} (max {np.max(t3 - t2):5.1f}) | to next tile {np.nanmean(nxt - t3) if np.isfinite(nxt).any() else float('nan'):5.2f} us")
    w = buf[1 << 19:(1 << 19) + nwg * 128].reshape(nwg, 8, 4, 4).astype(np.float64)
    steps = nk * np.array(per_wg, dtype=np.float64)[:, None, None]       # K-steps a workgroup ran
    print(f"   per K-step and phase, shader cycles in sections [issue, vmcnt, barrier+lgkm, mfma+barrier] (total {(w.sum((-1, -2)) / steps[:, :, 0]).mean():.0f}):")
    for P in range(4):
        g0, g1 = (w[:, :4, P] / steps).mean((0, 1)), (w[:, 4:, P] / steps).mean((0, 1))
        print(f"      phase {P}: waves 0-3 {g0.round(0)} = {g0.sum():.0f} | waves 4-7 {g1.round(0)} = {g1.sum():.0f}")

for name, M, N, Kd, epi, odt in [("dO none", T, 768, 768, 0, torch.bfloat16), ("qkv bias", T, 2304, 768, 1, torch.bfloat16),
                                  ("oproj resid f32", T, 768, 768, 3, torch.float32), ("fc1 gelu", T, 3072, 768, 2, torch.bfloat16),
                                  ("dU dgelu", T, 3072, 768, 4, torch.bfloat16), ("dU dgelu 8-bit stream", T, 3072, 768, 6, torch.bfloat16),
                                  ("dU shape, no epilogue", T, 3072, 768, 0, torch.bfloat16),
                                  ("dXn2 none K3072", T, 768, 3072, 0, torch.bfloat16), ("sq8192", 8192, 8192, 8192, 0, torch.bfloat16)]:
    run(M, N, Kd, epi, odt, name)
