#!/usr/bin/env python3
"""Whole training step under several builds of the library, in ONE process on ONE device, interleaved (guide rule 24):
box-to-box variance of bench.py is +-1.5 %, larger than most kernel-level changes.  GPU only; dev tool.

    python tools/step_ab.py base,product [rounds] [steps-per-round]      # env: arch=vit_b_16|vit_s_16|..., robust=1
A name may carry host-side switches: `product+bf16gelu` runs the product library with the bf16 gelu' stream of rounds 1 - 3
(encoder.GELU_STREAM_U8 = False).
"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _devlib
import bench as B
from noise_robust_vit_amd.train import TrainConfig, Trainer

libs = (sys.argv[1] if len(sys.argv) > 1 else "product").split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
arch = os.environ.get("arch", "vit_b_16")
robust = os.environ.get("robust", "") == "1"
dev = torch.device("cuda:0")
kind, kw = B.ARCHS[arch][0], B.ARCHS[arch][1]
batch = 128 if arch.endswith("l_16") else 256
model = B.build_model(arch, robust=robust).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=gen, device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), generator=gen, device=dev)

from noise_robust_vit_amd import encoder as _enc


def select(name):
    lib, *flags = name.split("+")
    _devlib.use_library(lib)
    _enc.GELU_STREAM_U8 = "bf16gelu" not in flags


def run(n):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        e[i].record(); trainer.step(x, y)
    e[n].record(); torch.cuda.synchronize()
    return [e[i].elapsed_time(e[i + 1]) for i in range(n)]

for l in libs:
    select(l); run(3)
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        select(l)
        run(1)
        res[l] += run(steps)
ref = statistics.median(res[libs[0]])
for l in libs:
    med = statistics.median(res[l])
    print(f"{arch}{' robust' if robust else ''} {l:10s}: median step {med:7.3f} ms  (min {min(res[l]):7.3f})  {batch / med * 1e3:8.1f} img/s  "
          f"{(med / ref - 1) * 100:+.2f} % vs {libs[0]}", flush=True)

# per-kernel-class device time inside the step (HIP events around every C-ABI launch), per library
from noise_robust_vit_amd import kernels as K
for l in libs:
    select(l)
    trainer.forward_backward(x, y)
    with K.LaunchProfile() as prof:
        for _ in range(3):
            trainer.forward_backward(x, y)
    summ = prof.summary()
    top = sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]
    print(f"{l:10s} in-step ms/step: " + "  ".join(f"{k} {v['ms'] / 3:.3f}" for k, v in top), flush=True)
