#!/usr/bin/env python3
"""cProfile of the host side of training steps (Python + ctypes + allocator), main thread and the autograd thread's share
through wall-clock of backward().  GPU only; dev tool.  env: arch=...   argv: steps"""
import os, sys, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from noise_robust_vit_amd.train import TrainConfig, Trainer
arch = os.environ.get("arch", "vit_s_16")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
kind, kw = B.ARCHS[arch][0], B.ARCHS[arch][1]
batch = 128 if arch.endswith("l_16") else 256
model = B.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), device=dev)
for _ in range(5): trainer.step(x, y)
torch.cuda.synchronize()
# the backward runs on autograd's thread: profile it by calling the Function's backward through the engine is not visible to
# cProfile of the main thread, so time phases by wall clock as well
t = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if trainer.reducer is not None: trainer.reducer.begin_step()
    loss = trainer.loss_fn(model(x), y) if kind != "mae" else model(x)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    if trainer.reducer is not None: trainer.reducer.finish_step()
    trainer.optimizer_step()
    t3 = time.perf_counter()
    t["fwd"] += t1 - t0; t["bwd"] += t2 - t1; t["opt"] += t3 - t2
print(f"{arch}: host ms/step with an EMPTY device queue (sync before every step): " + "  ".join(f"{k} {1e3 * v / steps:.2f}" for k, v in t.items()))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    trainer.step(x, y)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
