#!/usr/bin/env python3
"""Which Python call sites issue the small PyTorch fill / copy / add kernels of a training step?  (torch.profiler with stacks,
one step; the HIP path's own kernels are C-ABI calls and do not show up as aten ops.)  GPU only; dev tool.  env: arch=..."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench as B
from noise_robust_vit_amd.train import TrainConfig, Trainer
arch = os.environ.get("arch", "vit_b_16")
dev = torch.device("cuda:0")
kind, kw = B.ARCHS[arch][0], B.ARCHS[arch][1]
batch = 128 if arch.endswith("l_16") else 256
model = B.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), device=dev)
for _ in range(3): trainer.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    trainer.step(x, y)
    torch.cuda.synchronize()
want = ("aten::zero_", "aten::fill_", "aten::zeros", "aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::mul_", "aten::clone",
        "aten::contiguous", "aten::_to_copy", "aten::index", "aten::index_put_")
sites = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        frame = next((s for s in ev.stack if "/root/repo" in s or "noise_robust_vit_amd" in s or "bench.py" in s), ev.stack[0] if ev.stack else "?")
        sites[(ev.name, frame.strip()[-110:])] += 1
for (name, frame), n in sites.most_common(40):
    print(f"{n:5d}  {name:18s} {frame}")
