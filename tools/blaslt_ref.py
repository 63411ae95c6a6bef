#!/usr/bin/env python3
"""Known-good reference point (guide rule 10): hipBLASLt through torch.matmul on the ViT-B/16 GEMM shapes, same random data.
Dev tool, GPU only; not part of the product path."""
import torch
dev = torch.device("cuda:0")
T = 50432
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).bfloat16()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, M, N, K in [("qkv", T, 2304, 768), ("oproj/dO", T, 768, 768), ("fc1/dU", T, 3072, 768), ("fc2/dXn2", T, 768, 3072),
                      ("dXn1", T, 768, 2304), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]:
    A = rnd(M, K); B = rnd(N, K); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: torch.matmul(A, B.t(), out=out))
    print(f"blaslt NT {name:10s} M={M} N={N} K={K} {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TFLOP/s")
for name, M, N in [("dWo", 768, 768), ("dWqkv", 2304, 768), ("dW2", 768, 3072), ("dW1", 3072, 768)]:
    A = rnd(T, M); B = rnd(T, N); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: torch.matmul(A.t(), B, out=out))
    print(f"blaslt TN {name:10s} M={M} N={N} T={T} {ms:.3f} ms {2*M*N*T/ms/1e9:.1f} TFLOP/s")
