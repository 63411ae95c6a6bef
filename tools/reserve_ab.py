#!/usr/bin/env python3
"""What does leaving CUs to a collective cost the single-GPU step?  nrv_set_reserved_cus(n) for n in the list, interleaved rounds of
the whole training step in one process.  GPU only; dev tool.   python tools/reserve_ab.py [n ...]     env: arch=..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from noise_robust_vit_amd import kernels as K
from noise_robust_vit_amd.train import TrainConfig, Trainer
ns = [int(v) for v in sys.argv[1:]] or [0, 8, 16, 32]
arch = os.environ.get("arch", "vit_b_16")
dev = torch.device("cuda:0")
kind, kw = B.ARCHS[arch][0], B.ARCHS[arch][1]
batch = 128 if arch.endswith("l_16") else 256
model = B.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=g, device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), generator=g, device=dev)
def run(n):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        e[i].record(); trainer.step(x, y)
    e[n].record(); torch.cuda.synchronize()
    return [e[i].elapsed_time(e[i + 1]) for i in range(n)]
res = {n: [] for n in ns}
try:
    for n in ns:
        K.set_reserved_cus(n); run(3)
    for _ in range(5):
        for n in ns:
            K.set_reserved_cus(n); run(1); res[n] += run(6)
finally:
    K.set_reserved_cus(0)
ref = statistics.median(res[ns[0]])
for n in ns:
    m = statistics.median(res[n])
    print(f"{arch} reserved CUs {n:3d}: median step {m:7.3f} ms  {batch / m * 1e3:8.1f} img/s  {(m / ref - 1) * 100:+.2f} %", flush=True)
