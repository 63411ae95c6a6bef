#!/usr/bin/env python3
"""Attention forward / backward on the row-major and the blocked layouts of qkv / out (include/nrv.h NRV_ATTN_*_BLOCKED):
bit-equality of the results across layouts, then interleaved timing.  GPU only; dev tool.  env: B N H"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_robust_vit_amd import kernels as K

dev = torch.device("cuda:0")
B, N, H, dh = int(os.environ.get("B", 256)), int(os.environ.get("N", 197)), int(os.environ.get("H", 12)), 64
T = B * N
g = torch.Generator(device=dev).manual_seed(7)
qkv = (torch.randn(T, 3 * H * dh, generator=g, device=dev) * 0.5).bfloat16()
do = (torch.randn(T, H * dh, generator=g, device=dev) * 0.5).bfloat16()


def blk(t, nb):            # [T, nb*dh] -> [nb, T, dh]
    return t.reshape(T, nb, dh).permute(1, 0, 2).contiguous()


def unblk(t, nb):
    return t.permute(1, 0, 2).reshape(T, nb * dh).contiguous()


data = {}
for lay in (0, 1, 2, 3):
    q = blk(qkv, 3 * H) if lay & 1 else qkv
    d = blk(do, H) if lay & 2 else do
    o, lse = K.attn_fwd(q, B, N, H, dh, 0.125, layout=lay)
    dq = K.attn_bwd(q, o, d, lse, B, N, H, dh, 0.125, layout=lay)
    data[lay] = (q, d, o, lse, dq)
o0, lse0, dq0 = data[0][2], data[0][3], data[0][4]
for lay in (1, 2, 3):
    o = unblk(data[lay][2], H) if lay & 2 else data[lay][2]
    dq = unblk(data[lay][4], 3 * H) if lay & 1 else data[lay][4]
    print(f"layout {lay}: out bit-equal {torch.equal(o, o0)}  lse {torch.equal(data[lay][3], lse0)}  dqkv {torch.equal(dq, dq0)}")


def timeit(fn, n=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


cold = os.environ.get("cold", "1") == "1"
junk = torch.empty(256 * 1024 * 1024 // 4, device=dev)       # evict the Infinity Cache between calls: in the step the operands are cold
res = {(lay, w): [] for lay in (0, 1, 2, 3) for w in ("fwd", "bwd")}
for _ in range(7):
    for lay in (0, 1, 2, 3):
        q, d, o, lse, _ = data[lay]
        for w, fn in (("fwd", lambda: K.attn_fwd(q, B, N, H, dh, 0.125, layout=lay)),
                      ("bwd", lambda: K.attn_bwd(q, o, d, lse, B, N, H, dh, 0.125, layout=lay))):
            if cold:
                ts = []
                for _ in range(3):
                    junk.add_(1.0)
                    ts.append(timeit(fn, 1))
                res[(lay, w)].append(statistics.median(ts))
            else:
                res[(lay, w)].append(timeit(fn))
names = {0: "row-major", 1: "qkv blocked", 2: "out blocked", 3: "qkv + out blocked"}
algf, algb = 2 * T * H * dh * 4, 2 * T * H * dh * 8
for lay in (0, 1, 2, 3):
    f, b = statistics.median(res[(lay, "fwd")]), statistics.median(res[(lay, "bwd")])
    print(f"B {B} N {N} H {H} {'cold' if cold else 'warm'} | {names[lay]:18s} fwd {f:7.1f} us ({algf / f / 1e6:5.2f} TB/s)   bwd (dq + dkv) {b:7.1f} us ({algb / b / 1e6:5.2f} TB/s)")
