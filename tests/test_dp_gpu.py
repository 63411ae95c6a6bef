"""GPU: the N > 1 step (gradient sink + bucketed async all-reduce + clip + AdamW) through the real HIP encoder path.

A 1-GPU box cannot run RCCL with two ranks on one device, so the two ranks share cuda:0 and exchange gradients over
`gloo` (host-staged); everything else -- the sink writing dW into the flat buffer from inside the hand-scheduled
backward, bucket readiness, averaging, the optimizer reading the views -- is the production code path.
Checked against a single-process run on the concatenated batch.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from noise_robust_vit_amd import VisionTransformer
from noise_robust_vit_amd.parallel import GradReducer
from noise_robust_vit_amd.train import Trainer, TrainConfig
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
m = VisionTransformer(image_size=32, patch_size=16, num_layers=2, num_heads=3, hidden_dim=192, mlp_dim=768, num_classes=10)
torch.nn.init.normal_(m.heads.head.weight, std=0.02)
m = m.to(dev).train()
red = GradReducer(m, world, bucket_mib=0.5) if world > 1 else None
tr = Trainer(m, TrainConfig(lr=1e-3, grad_max_norm=5.0), red)
g = torch.Generator().manual_seed(7)
X = torch.randn(8, 3, 32, 32, generator=g); Y = torch.randint(0, 10, (8,), generator=g)
per = 8 // world
x, y = X[rank*per:(rank+1)*per].to(dev), Y[rank*per:(rank+1)*per].to(dev)
losses = [tr.step(x, y).item() for _ in range(3)]
tr.forward_backward(x, y)
torch.cuda.synchronize()
if rank == 0:
    out = {"loss": losses, "nbuckets": len(red.buckets) if red else 0}
    for k, p in m.named_parameters():
        out["g." + k] = p.grad.detach().float().cpu().reshape(-1)[:64].tolist()
        out["gn." + k] = p.grad.detach().float().norm().item()
        out["w." + k] = p.detach().float().cpu().reshape(-1)[:64].tolist()
    json.dump(out, open(sys.argv[2], "w"))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
"""


def _run(world, out_path, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out_path)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0


def test_two_ranks_on_one_gpu_match_single_process(dev, tmp_path):
    import json
    _run(1, tmp_path / "one.json", tmp_path)
    _run(2, tmp_path / "two.json", tmp_path)
    one = json.load(open(tmp_path / "one.json")); two = json.load(open(tmp_path / "two.json"))
    assert two["nbuckets"] > 1
    # per-rank means of half batches, averaged == full-batch mean (CE is a mean over samples)
    for a, b in zip(one["loss"], two["loss"]):
        assert abs(a - b) < 0.25          # rank 0 reports ITS shard's loss; only sanity here
    for k in one:
        if k.startswith("gn."):
            assert abs(one[k] - two[k]) <= 2e-2 * max(one[k], 1e-6) + 1e-6, (k, one[k], two[k])
        if k.startswith("g.") or k.startswith("w."):
            a, b = torch.tensor(one[k]), torch.tensor(two[k])
            assert (a - b).abs().max() <= 2e-2 * a.abs().max() + 2e-5, k
