#!/usr/bin/env python3
"""Which Python call sites issue PyTorch (aten) device ops in one training step?  The HIP path's own kernels are C-ABI calls
and do not show up; what does is the glue around them (fills, copies, adds, index bookkeeping).  A TorchDispatchMode records
every aten op with the innermost frame of this repository on its stack; autograd runs single-threaded so that the backward
is seen too.  GPU only; dev tool.  env: arch=..."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench as B
from noise_robust_vit_amd.train import TrainConfig, Trainer

arch = os.environ.get("arch", "vit_b_16")
dev = torch.device("cuda:0")
kind, kw = B.ARCHS[arch][0], B.ARCHS[arch][1]
batch = int(os.environ.get("batch", 128 if arch.endswith("l_16") else 256))
model = B.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), device=dev)
for _ in range(3):
    trainer.step(x, y)
torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOKERNEL = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.slice",
            "aten.select", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.empty", "aten.as_strided", "aten.set_",
            "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.unbind", "aten.split", "aten.lift_fresh", "aten._local_scalar")
sites = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(NOKERNEL):
            frame = "?"
            for fs in reversed(traceback.extract_stack()):
                if fs.filename.startswith(ROOT) and "torch_op_sites" not in fs.filename:
                    frame = f"{os.path.relpath(fs.filename, ROOT)}:{fs.lineno} {fs.name}"
                    break
            sites[(name, frame)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
with Rec():
    trainer.step(x, y)
torch.cuda.synchronize()
print(f"{arch} batch {batch}: {sum(sites.values())} aten ops with device work in one step")
for (name, frame), n in sites.most_common(70):
    print(f"{n:5d}  {name:34s} {frame}")
