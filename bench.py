#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ViT-B/16 224 px bf16 training step on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch resident in HBM:
forward + cross-entropy(label_smoothing=0.1) + backward (+ RCCL gradient all-reduce when N > 1)
+ grad-clip + AdamW step (the reference harness step, examples/CIFAR100.py:115-141,90-97,191-192).
Weak scaling: the per-GPU batch is fixed (256), the global batch grows with N.

Rank 0 prints ONE JSON line (see the keys at the bottom).  `roofline` is measured live with HIP events around
every launch of the dominant kernel (the MFMA bf16 "NT" GEMM) during extra instrumented steps; `cpu_baseline`
times the CPU oracle (a port of the reference's forward + autograd backward) on a bounded sample, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARCHS = {
    # name: (kind, kwargs, fwd+bwd GFLOP per image)
    "vit_b_16": ("vt", dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072), 105.383),
    "vit_s_16": ("vt", dict(image_size=224, patch_size=16, num_layers=12, num_heads=6, hidden_dim=384, mlp_dim=1536), 27.593),
    "vit_l_16": ("vt", dict(image_size=224, patch_size=16, num_layers=24, num_heads=16, hidden_dim=1024, mlp_dim=4096), 369.328),
    "simplevit_b_16": ("simple", dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072), 104.830),
    "simplevit_s_16": ("simple", dict(image_size=224, patch_size=16, dim=384, depth=12, heads=6, mlp_dim=1536), 27.444),
    "simplevit_l_16": ("simple", dict(image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=4096), 367.400),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0


def build_model(arch: str, num_classes: int = 1000):
    from noise_robust_vit_amd import SimpleViT, VisionTransformer
    kind, kw, _ = ARCHS[arch]
    torch.manual_seed(0)
    if kind == "vt":
        m = VisionTransformer(num_classes=num_classes, **kw)
        # the reference zero-initialises heads.head (vit.py:304-306): re-randomise so logits/grads are non-trivial
        torch.nn.init.normal_(m.heads.head.weight, std=0.02)
    else:
        m = SimpleViT(num_classes=num_classes, **kw)
    return m


def pmc_traffic(kernel_class: str):
    """HBM/fabric bytes per launch of a kernel class from the committed PMC passes of this same command
    (profiles/r01_traffic_per_launch.json, produced by tools/traffic_from_pmc.py from separate `rocprofv3 --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE` runs of bench.py, with the gfx950 FETCH_SIZE x2 correction).  A profiler cannot run inside the
    timed process, so the number is read from that file; null when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_per_launch.json")
    try:
        with open(path) as f:
            return json.load(f)["per_launch_bytes"][kernel_class]["total"]
    except Exception:
        return None


def cpu_baseline(arch: str, seconds_budget: float = 12.0):
    """Reference-algorithm CPU path (oracle port: fp32 forward + autograd backward + CE) on the host cores: a bounded sample
    of the same workload -- batch 4 steps repeated for ~12 s of CPU work."""
    from oracle import simple_vit_oracle as SO
    from oracle import vit_oracle as VO
    kind, kw, _ = ARCHS[arch]
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)                   # a 1-GPU box's CPU share is 16 cores, whatever nproc reports
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    batch = 4
    if kind == "vt":
        sd = VO.vit_init_state_dict(seed=0, num_classes=1000, **kw)
        fwd = lambda s, x: VO.vit_forward(s, x, patch_size=kw["patch_size"], num_heads=kw["num_heads"])
    else:
        m = build_model(arch)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        fwd = lambda s, x: SO.simple_vit_forward(s, x, patch_size=kw["patch_size"], heads=kw["heads"])
    x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=g)
    y = torch.randint(0, 1000, (batch,), generator=g)

    def step():
        leaves = {k: v.detach().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        loss = SO.cross_entropy_ls(fwd(leaves, x), y)
        loss.backward()

    step()                                   # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 64:
            break
    return {"value": round(batch * n / el, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{arch} fp32 forward+CE+backward (oracle/), batch {batch}, {n} timed steps after 1 warm-up, "
                      f"{torch.get_num_threads()} torch threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="vit_b_16", choices=sorted(ARCHS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 256; 128 for *_l_16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel-class time table to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NRV_DIST_BACKEND", "nccl")      # "gloo" lets several ranks share one GPU (functional test)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from noise_robust_vit_amd import kernels as K
    from noise_robust_vit_amd.parallel import GradReducer
    from noise_robust_vit_amd.train import TrainConfig, Trainer

    batch = args.batch or (128 if args.arch.endswith("l_16") else 256)
    model = build_model(args.arch).to(dev).train()
    reducer = GradReducer(model, world) if world > 1 else None
    trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), reducer)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    kw = ARCHS[args.arch][1]
    x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=gen, device=dev, dtype=torch.float32).to(torch.bfloat16)
    y = torch.randint(0, 1000, (batch,), generator=gen, device=dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x, y)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x, y)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = batch * world * args.steps / elapsed
    gflop_img = ARCHS[args.arch][2]
    out = {
        "metric": "images/sec (fwd+bwd+optimizer) ViT-B/16 224px bf16" if args.arch == "vit_b_16"
                  else f"images/sec (fwd+bwd+optimizer) {args.arch} 224px bf16",
        "value": round(images_per_s, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.arch} 224px training step (BASELINE.json configs[2]: ViT-B/16 batch 256/GPU)"
                               if args.arch == "vit_b_16" else f"{args.arch} 224px training step",
                   "arch": args.arch, "per_gpu_batch": batch, "global_batch": batch * world, "tokens": 197 if ARCHS[args.arch][0] == "vt" else 196,
                   "parallelism": f"dp{world}", "step": "fwd + CE(ls=0.1) + bwd + allreduce + clip(5.0) + AdamW",
                   "residual_stream": "fp32", "gemm_operands": "bf16", "accumulate": "fp32"},
        "images_per_sec_per_gpu": round(images_per_s / world, 2),
        "step_tflops_per_gpu": round(images_per_s / world * gflop_img / 1e3, 1),
        "step_mfma_frac": round(images_per_s / world * gflop_img / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4),
        "loss": round(float(loss), 4),
    }

    if rank == 0 and not args.no_roofline:
        # instrumented steps: HIP events around every C-ABI launch on the compute stream
        with K.LaunchProfile() as prof:
            for _ in range(2):
                if world == 1:
                    trainer.forward_backward(x, y)      # gradients are overwritten in the flat buffer every step
        summ = prof.summary() if world == 1 else {}
        if "gemm_nt" in summ:
            gnt = summ["gemm_nt"]
            ach = gnt["flops"] / (gnt["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_kernel (MFMA bf16 NT GEMM, all epilogues)",
                               "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": pmc_traffic("gemm_nt"),
                               "launches_per_step": gnt["launches"] // 2,
                               "avg_launch_ms": round(gnt["ms"] / gnt["launches"], 4),
                               "alg_flop_per_launch": round(gnt["flops"] / gnt["launches"]),
                               "share_of_step_ms": round(gnt["ms"] / 2, 3)}
            out["kernel_ms_per_step"] = {k: round(v["ms"] / 2, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
            if args.breakdown:
                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
                    gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
                    print(f"  {k:16s} {v['launches']//2:4d} launches/step {v['ms']/2:8.3f} ms/step {tf:8.1f} TFLOP/s {gb:8.1f} GB/s(alg)",
                          file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.arch)
        except Exception as e:       # the baseline leg must never take the bench line down
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
