#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ViT-B/16 224 px bf16 training step on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch resident in HBM:
forward + cross-entropy(label_smoothing=0.1) + backward (+ RCCL gradient all-reduce when N > 1)
+ grad-clip + AdamW step (the reference harness step, examples/CIFAR100.py:115-141,90-97,191-192).
Weak scaling: the per-GPU batch is fixed (256), the global batch grows with N.

Protocol: W = 10 warm-up steps, then exactly K = 50 steps timed between barrier + device
synchronisation on both sides (`value`, `ms_per_step`: the contract's whole-job throughput, MAX over ranks); the same
K steps are also bracketed one by one with HIP events on the compute stream and the MEDIAN step is reported beside it
(`ms_per_step_median`).  Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every launch
of the dominant kernel (the MFMA bf16 "NT" GEMM) during two extra instrumented steps; `cpu_baseline` times the CPU
oracle (a port of the reference's forward + autograd backward) on a bounded sample, N = 1 only.

Other workloads of BASELINE.json (`--arch`): vit_s_16 (configs[1]), vit_l_16 (configs[3], batch 128), mae_b_16
(configs[4]: MAE over a ViT-B/16 encoder, 75 % mask -> 49 encoder tokens), the SimpleViT family; `--robust` selects the
Sinkhorn attention (the reference's `robust=True`), `--noise-std S` the noisy-input training of examples/nowak.py:153.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ROUND = "r04"


def _vit(L, H, D, M):
    return dict(image_size=224, patch_size=16, num_layers=L, num_heads=H, hidden_dim=D, mlp_dim=M)


def _simple(L, H, D, M):
    return dict(image_size=224, patch_size=16, dim=D, depth=L, heads=H, mlp_dim=M)


ARCHS = {
    # name: (kind, kwargs, fwd+bwd GFLOP per image)
    "vit_b_16": ("vt", _vit(12, 12, 768, 3072), 105.383),
    "vit_s_16": ("vt", _vit(12, 6, 384, 1536), 27.593),
    "vit_l_16": ("vt", _vit(24, 16, 1024, 4096), 369.328),
    "simplevit_b_16": ("simple", _simple(12, 12, 768, 3072), 104.830),
    "simplevit_s_16": ("simple", _simple(12, 6, 384, 1536), 27.444),
    "simplevit_l_16": ("simple", _simple(24, 16, 1024, 4096), 367.400),
    # MAE (mae.py:9-49): lucidrains-style ViT-B/16 encoder on the 49 kept tokens + the wrapper's default decoder (512 wide, 1 layer,
    # 8 heads) on all 196.  30.333 = BASELINE.md's encoder-only 25.935 (patch embedding on 196 tokens + 12 blocks on 49) + the
    # work the step also does: enc_to_dec 49 x 768 x 512, one decoder block at 196 tokens (qkv 512 -> 1536, out 512 -> 512,
    # QK^T / PV 8 heads x 64, MLP 512 -> 2048 -> 512) and to_pixels 147 x 512 x 768 = 1.465 GFLOP forward, x 3 for the step
    "mae_b_16": ("mae", dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072), 30.333),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0


def build_model(arch: str, num_classes: int = 1000, robust: bool = False):
    from noise_robust_vit_amd import SimpleViT, VisionTransformer
    kind, kw, _ = ARCHS[arch]
    torch.manual_seed(0)
    if kind == "vt":
        m = VisionTransformer(num_classes=num_classes, robust=robust, **kw)
        # the reference zero-initialises heads.head (vit.py:304-306): re-randomise so logits/grads are non-trivial
        torch.nn.init.normal_(m.heads.head.weight, std=0.02)
    elif kind == "simple":
        m = SimpleViT(num_classes=num_classes, robust=robust, **kw)
    else:
        from noise_robust_vit_amd.lucid_vit import ViT
        from noise_robust_vit_amd.mae import MAE
        if robust:
            raise SystemExit("--robust: the MAE encoder family (learnable_memory_vit.py) has no Sinkhorn switch in the reference")
        m = MAE(encoder=ViT(**kw), decoder_dim=512, masking_ratio=0.75, decoder_depth=1, decoder_heads=8, decoder_dim_head=64)
    return m


def pmc_traffic(arch: str, batch: int, kernel_class: str):
    """HBM/fabric bytes per launch of a kernel class from the PMC passes committed THIS round for THIS workload
    (profiles/<round>_traffic_per_launch_<arch>_b<batch>.json, produced by tools/traffic_from_pmc.py from separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this command, gfx950 FETCH_SIZE x2 correction applied).
    A profiler cannot run inside the timed process: the number is read from that file and is None (with the reason in
    `traffic_source`) when no file matches -- never another workload's or another round's measurement."""
    name = f"{ROUND}_traffic_per_launch_{arch}_b{batch}.json"
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f)["per_launch_bytes"][kernel_class]["total"], "profiles/" + name
    except Exception:
        return None, f"no profiles/{name}: PMC passes not collected for this workload in {ROUND}"


def host_cores():
    """(threads to use, how that was derived): the CPUs this process may run on, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
        how = "sched_getaffinity"
    except Exception:
        n, how = os.cpu_count() or 1, "cpu_count"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                n, how = q, "cgroup cpu.max"
    except Exception:
        pass
    return n, how


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(arch: str, robust: bool, budget_s: float = 24.0):
    """Reference-algorithm CPU path (oracle port: forward + CE(ls 0.1) + autograd backward) on the host cores, as
    BASELINE.md §4 prescribes: batch 16 (8 for L), fp32 and bf16, 1 warm-up + up to 5 timed steps each (median), bounded
    to ~`budget_s` seconds of CPU work in total (a bounded sample of the same workload)."""
    from oracle import simple_vit_oracle as SO
    from oracle import vit_oracle as VO
    kind, kw, _ = ARCHS[arch]
    cores, how = host_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    batch = 8 if arch.endswith("l_16") else 16
    if kind == "mae":
        from oracle import mae_oracle as MO
        m = build_model(arch)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        idx = torch.rand(batch, 196, generator=g).argsort(dim=-1)
    elif kind == "vt":
        sd = VO.vit_init_state_dict(seed=0, num_classes=1000, **kw)
    else:
        m = build_model(arch, robust=robust)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x32 = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=g)
    y = torch.randint(0, 1000, (batch,), generator=g)

    def loss_of(leaves, x):
        if kind == "mae":
            return MO.mae_forward(leaves, x, idx, patch_size=16, enc_heads=kw["heads"], dec_heads=8)
        if kind == "vt":
            logits = VO.vit_forward(leaves, x, patch_size=kw["patch_size"], num_heads=kw["num_heads"], robust=robust)
        else:
            logits = SO.simple_vit_forward(leaves, x, patch_size=kw["patch_size"], heads=kw["heads"], robust=robust)
        return SO.cross_entropy_ls(logits.float(), y)

    def leg(dtype, seconds):
        sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        x = x32.to(dtype)

        def step():
            leaves = {k: v.detach().requires_grad_(v.is_floating_point()) for k, v in sdd.items()}
            loss_of(leaves, x).backward()

        step()                               # warm-up (allocator, thread pool)
        ts, t_start = [], time.perf_counter()
        while len(ts) < 5 and (len(ts) < 2 or time.perf_counter() - t_start < seconds):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        return batch / statistics.median(ts), len(ts)

    v32, n32 = leg(torch.float32, budget_s * 0.6)
    v16, n16 = leg(torch.bfloat16, budget_s * 0.4)
    return {"value": round(v32, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "bf16_value": round(v16, 3), "cpu_model": cpu_model(), "torch_threads": torch.get_num_threads(), "cores_from": how,
            "sample": f"{arch} forward+loss+backward of oracle/ (port of the reference CPU path), batch {batch}: fp32 median of "
                      f"{n32} steps, bf16 median of {n16} steps, 1 warm-up each"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="vit_b_16", choices=sorted(ARCHS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 256; 128 for *_l_16)")
    ap.add_argument("--robust", action="store_true", help="Sinkhorn attention (the reference's robust=True)")
    ap.add_argument("--noise-std", type=float, default=0.0, help="noisy-input training, x + N(0, s^2) (examples/nowak.py:153)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel-class time table to stderr")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"], help="gradient exchange precision (N > 1)")
    ap.add_argument("--bucket-mib", type=float, default=64.0, help="all-reduce bucket size")
    ap.add_argument("--tail-mib", type=float, default=8.0, help="cap of the last (non-overlappable) bucket")
    ap.add_argument("--rccl-max-ctas", type=int, default=0,
                    help="CU budget of RCCL's kernels (0: RCCL's default -- what runs at N > 1 until an N > 1 A/B exists; 8 was reasoned for "
                         "the 605 MB a ViT-B/16 step exchanges per GPU, never measured)")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="CUs the GEMM launches leave free for the collective's kernels when N > 1 (0: none, the default; unmeasured at N > 1)")
    ap.add_argument("--wgrad", default="default", choices=["default", "grouped", "single"],
                    help="weight gradients: one grouped stream-K launch per layer, or one split-K launch per gradient (A/B switch)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (Trainer.capture)")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ          # started by torch.distributed.run
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
    import torch.distributed as dist
    from noise_robust_vit_amd import kernels as K
    from noise_robust_vit_amd.parallel import GradReducer, make_process_group
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    use_pg = world > 1 or launched
    pg_desc = ""
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        args.rccl_max_ctas = max(args.rccl_max_ctas, 0)
        args.reserve_cus = max(args.reserve_cus, 0) if world > 1 else 0
        pg_desc = make_process_group(rank, world, device=dev, backend="nccl", max_ctas=args.rccl_max_ctas)   # "nccl" IS RCCL on ROCm

    if args.wgrad != "default":
        from noise_robust_vit_amd import encoder as _enc
        _enc.WGRAD_GROUPED = args.wgrad == "grouped"
    kind = ARCHS[args.arch][0]
    batch = args.batch or (128 if args.arch.endswith("l_16") else 256)
    model = build_model(args.arch, robust=args.robust).to(dev).train()
    # under the launcher the collectives are issued at any world size (world 1: a one-rank RCCL group, same calls)
    reducer = GradReducer(model, world, force_collectives=world == 1, bucket_mib=args.bucket_mib, tail_mib=args.tail_mib,
                          grad_dtype=args.grad_dtype, reserve_cus=max(args.reserve_cus, 0)) if use_pg else None
    compute_loss = (lambda m, xb, yb: m(xb)) if kind == "mae" else None
    trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0, noise_std=args.noise_std), reducer,
                      compute_loss=compute_loss)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    kw = ARCHS[args.arch][1]
    x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=gen, device=dev, dtype=torch.float32).to(torch.bfloat16)
    y = torch.randint(0, 1000, (batch,), generator=gen, device=dev)

    def sync():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x, y)
    if args.graph:
        if use_pg:
            raise SystemExit("--graph: single process only (collectives are not captured)")
        trainer.capture(x, y)
        for _ in range(2):
            trainer.step(x, y)
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()                        # on the compute stream, between steps: per-step device time
        loss = trainer.step(x, y)
    marks[args.steps].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    med = statistics.median(per_step)
    if world > 1:
        t = torch.tensor([elapsed, med], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, med = t[0].item(), t[1].item()

    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = batch * world * args.steps / elapsed
    gflop_img = ARCHS[args.arch][2]
    tokens = {"vt": 197, "simple": 196, "mae": 49}[kind]
    headline = args.arch == "vit_b_16" and not args.robust and args.noise_std == 0.0
    variant = args.arch + (" robust(Sinkhorn)" if args.robust else "") + (f" noise_std={args.noise_std}" if args.noise_std else "")
    out = {
        "metric": "images/sec (fwd+bwd+optimizer) ViT-B/16 224px bf16" if headline
                  else f"images/sec (fwd+bwd+optimizer) {variant} 224px bf16",
        "value": round(images_per_s, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{variant} 224px training step"
                               + (" (BASELINE.json configs[2]: ViT-B/16 batch 256/GPU)" if headline else ""),
                   "arch": args.arch, "robust": args.robust, "noise_std": args.noise_std,
                   "per_gpu_batch": batch, "global_batch": batch * world, "tokens": tokens,
                   "parallelism": f"dp{world}",
                   "step": ("fwd + MSE on masked patches" if kind == "mae" else "fwd + CE(ls=0.1)") + " + bwd + allreduce + clip(5.0) + AdamW",
                   "collectives": ("none (single process, no process group)" if not use_pg else
                                   f"rccl all_reduce(AVG) of {args.grad_dtype} gradients per {args.bucket_mib:g} MiB bucket (last bucket <= "
                                   f"{args.tail_mib:g} MiB; buckets {[round(b / 2**20, 1) for b in reducer.bucket_bytes()]} MiB); {pg_desc}; "
                                   f"GEMM launches planned for the device's CUs minus {reducer.reserve_cus}"
                                   + (", forced at world 1" if world == 1 else ", overlapped with backward")),
                   "launch": "one HIP graph per step (Trainer.capture)" if args.graph else "eager (one C-ABI call per kernel)",
                   "streams": "the layer's grouped weight-gradient launch goes to a second HIP stream where the dX grid leaves >= 15 % of its "
                              "CU-rounds idle (encoder.WGRAD_STREAM = auto; not on ViT-B/16 at batch 256); the per-launch timing leg runs on one stream",
                   "residual_stream": "fp32", "gemm_operands": "bf16", "accumulate": "fp32",
                   "timing": f"{args.warmup} warm-up + {args.steps} timed steps, barrier + synchronize on both sides; "
                             "ms_per_step_median = median of per-step HIP-event spans on the compute stream"},
        "ms_per_step_median": round(med, 3),
        "images_per_sec_from_median": round(batch * world / med * 1e3, 2),
        "images_per_sec_per_gpu": round(images_per_s / world, 2),
        "step_tflops_per_gpu": round(images_per_s / world * gflop_img / 1e3, 1),
        "step_mfma_frac": round(images_per_s / world * gflop_img / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4),
        "gflop_per_image": gflop_img,
        "loss": round(float(loss), 4),
    }

    if rank == 0 and not args.no_roofline and world == 1:
        # instrumented steps: HIP events around every C-ABI launch on the compute stream
        with K.LaunchProfile() as prof:
            for _ in range(2):
                trainer.forward_backward(x, y)      # gradients are overwritten in the flat buffer every step
        summ = prof.summary()
        if "gemm_nt" in summ:
            gnt = summ["gemm_nt"]
            ach = gnt["flops"] / (gnt["ms"] * 1e-3) / 1e12
            traffic, tsrc = pmc_traffic(args.arch, batch, "gemm_nt")
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt8_kernel / gemm_nt_kernel (MFMA bf16 NT GEMM, phased K loop, all epilogues)",
                               "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": tsrc,
                               "launches_per_step": gnt["launches"] // 2,
                               "avg_launch_ms": round(gnt["ms"] / gnt["launches"], 4),
                               "alg_flop_per_launch": round(gnt["flops"] / gnt["launches"]),
                               "share_of_step_ms": round(gnt["ms"] / 2, 3)}
            out["kernel_ms_per_step"] = {k: round(v["ms"] / 2, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
            # the HBM-bound kernel classes against the 8 TB/s roofline (reported separately)
            out["hbm_bound_kernels"] = {k: {"alg_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                            "frac_of_8TBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
                                        for k, v in summ.items() if v["flops"] == 0 and v["bytes"] > 0 and v["ms"] > 0}
            if args.breakdown:
                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
                    gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
                    print(f"  {k:16s} {v['launches']//2:4d} launches/step {v['ms']/2:8.3f} ms/step {tf:8.1f} TFLOP/s {gb:8.1f} GB/s(alg)",
                          file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.arch, args.robust)
        except Exception as e:       # the baseline leg must never take the bench line down
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
