"""Capture one training step into a HIP graph WITHOUT replaying it and list what the graph holds (hipGraphDebugDotPrint):
node kinds, kernel names, and every memcpy / memset node with its operands -- a host pointer in a memcpy node is read again at
every replay, long after the host buffer is gone.  Usage: python tools/graph_nodes.py ARCH [OUT.dot]"""
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from noise_robust_vit_amd.train import TrainConfig, Trainer  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "mae_b_16"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"graph_{arch}.dot")
batch = int(os.environ.get("batch", "64"))
dev = torch.device("cuda", 0)
kind, kw, _ = bench.ARCHS[arch]
model = bench.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(batch, 3, 224, 224, generator=g, device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), generator=g, device=dev)
for _ in range(2):
    trainer.step(x, y)
torch.cuda.synchronize()
os.makedirs(os.path.dirname(out), exist_ok=True)
trainer.capture(x, y, debug_dump=out, _allow_custom_loss=True)
trainer._graph = None                   # never replayed here
torch.cuda.synchronize()
txt = open(out).read()
print(f"{arch}: dot file {len(txt)} bytes")
labels = re.findall(r'label="([^"]*)"', txt)
kinds = collections.Counter()
for lab in labels:
    first = lab.split("\\n")[0].split("\n")[0]
    kinds[re.sub(r"[0-9]+$", "", first.strip())[:80]] += 1
for k, v in kinds.most_common(60):
    print(f"  {v:5d}  {k}")
print("--- memcpy / memset / host nodes in full")
for lab in labels:
    low = lab.lower()
    if "memcpy" in low or "memset" in low or "host" in low:
        print("  ", lab.replace("\\n", " | ")[:400])
