#!/usr/bin/env python3
"""Attention backward of two builds on the same data: dq / dk / dv against an fp32 torch reference and against each other, then
interleaved timing.  GPU only; dev tool.    python tools/attn_ab.py LIB_A LIB_B [N ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _devlib
from noise_robust_vit_amd import kernels as K
dev = torch.device("cuda:0")
la, lb = sys.argv[1], sys.argv[2]
Ns = [int(v) for v in sys.argv[3:]] or [197, 196, 208, 193, 224]
B, H, dh = int(os.environ.get("B", 256)), 12, 64
def timeit(fn, n=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for N in Ns:
    g = torch.Generator(device=dev).manual_seed(N)
    qkv = (torch.randn(B * N, 3 * H * dh, generator=g, device=dev) * 0.5).bfloat16()
    do = (torch.randn(B * N, H * dh, generator=g, device=dev) * 0.5).bfloat16()
    _devlib.use_library(la)
    o, lse = K.attn_fwd(qkv, B, N, H, dh, 0.125)
    outs = {}
    for l in (la, lb):
        _devlib.use_library(l)
        outs[l] = [K.attn_bwd(qkv, o, do, lse, B, N, H, dh, 0.125).clone() for _ in range(3)]
    # fp32 reference on a slice of the batch
    nb = 4
    q, k, v = (t.reshape(B, N, H, dh)[:nb].permute(0, 2, 1, 3).float() for t in qkv.reshape(B, N, 3, H * dh).unbind(2))
    dO = do.reshape(B, N, H, dh)[:nb].permute(0, 2, 1, 3).float()
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    P = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1)
    (P @ v).backward(dO)
    ref = torch.stack([t.grad.permute(0, 2, 1, 3).reshape(nb, N, H * dh) for t in (q, k, v)], 2).reshape(nb * N, 3 * H * dh)
    for l in (la, lb):
        got = outs[l][0][:nb * N].float()
        err = [float((got.reshape(nb * N, 3, -1)[:, i] - ref.reshape(nb * N, 3, -1)[:, i]).norm() / ref.reshape(nb * N, 3, -1)[:, i].norm()) for i in range(3)]
        same = all(torch.equal(outs[l][0], x) for x in outs[l][1:])
        print(f"N {N} {l:8s}: rel-L2 vs fp32 torch dq {err[0]:.2e} dk {err[1]:.2e} dv {err[2]:.2e}; repeat-identical {same}; finite {bool(torch.isfinite(outs[l][0]).all())}")
    d = (outs[la][0].float() - outs[lb][0].float()).abs().max().item()
    print(f"N {N}: max |{la} - {lb}| = {d:.3e} (max |value| {outs[la][0].float().abs().max().item():.2f})")
    res = {la: [], lb: []}
    for _ in range(5):
        for l in (la, lb):
            _devlib.use_library(l)
            res[l].append(timeit(lambda: K.attn_bwd(qkv, o, do, lse, B, N, H, dh, 0.125)))
    print(f"N {N}: " + "  ".join(f"{l} {statistics.median(res[l]):7.1f} us" for l in (la, lb)), flush=True)
