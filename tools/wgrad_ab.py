#!/usr/bin/env python3
"""Whole training step with the weight gradients on the main stream, on a second HIP stream, or chosen per GEMM (encoder.WGRAD_STREAM),
interleaved in one process.  GPU only; dev tool.     python tools/wgrad_ab.py [rounds] [steps]     env: arch=..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from noise_robust_vit_amd import encoder
from noise_robust_vit_amd.train import TrainConfig, Trainer

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
arch = os.environ.get("arch", "vit_b_16")
dev = torch.device("cuda:0")
kind, kw = B.ARCHS[arch][0], B.ARCHS[arch][1]
batch = 128 if arch.endswith("l_16") else 256
model = B.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=gen, device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), generator=gen, device=dev)

def run(n):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        e[i].record(); trainer.step(x, y)
    e[n].record(); torch.cuda.synchronize()
    return [e[i].elapsed_time(e[i + 1]) for i in range(n)]

modes = [("one stream", False), ("wgrad stream", True), ("auto", "auto")]
for _, m in modes:
    encoder.WGRAD_STREAM = m; run(3)
res = {n: [] for n, _ in modes}
for _ in range(rounds):
    for n, m in modes:
        encoder.WGRAD_STREAM = m
        run(1)
        res[n] += run(steps)
ref = statistics.median(res[modes[0][0]])
for n, _ in modes:
    med = statistics.median(res[n])
    print(f"{arch} {n:14s}: median step {med:7.3f} ms (min {min(res[n]):7.3f})  {batch / med * 1e3:8.1f} img/s  {(med / ref - 1) * 100:+.2f} %", flush=True)
# same gradients either way?
outs = []
for n, m in modes:
    encoder.WGRAD_STREAM = m
    torch.manual_seed(7)                    # MAE draws its mask from the global generator
    trainer.forward_backward(x, y); torch.cuda.synchronize()
    outs.append(trainer.reducer.flat.clone())
print("flat gradient buffers bit-equal:", all(torch.equal(outs[0], o) for o in outs[1:]))
