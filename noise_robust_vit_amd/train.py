"""Training-step harness around the HIP encoder path (counterpart of the reference's examples/CIFAR100.py and
examples/baseline.py, whose trainer base class `omega.Trainer` is not in the reference tree -- SURVEY.md §8a row a15).

What the reference harness does per step and this module reproduces (file:line in /root/reference/examples/):
  * even batch split across ranks                      CIFAR100.py:22, baseline.py:16
  * loss = cross_entropy(model(x), y, label_smoothing=0.1)   CIFAR100.py:139, baseline.py:70
  * optional noisy input  x + N(0, sigma^2)            nowak.py:152-159
  * gradient clipping, max-norm 5.0                    CIFAR100.py:192, baseline.py:127
  * AdamW(lr, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999))   CIFAR100.py:90-97,191
  * per-iteration LinearLR(1e-3 -> 1) warm-up over 10 % of the epochs, then cosine to 0.05 lr   CIFAR100.py:99-113,165-166
  * DDP: gradients averaged over ranks (all-reduce SUM / world) before the optimizer step

On the MI355X the clip + AdamW arithmetic runs in `optim.FusedAdamW` (two HBM-bound HIP kernels over flat buffers,
SURVEY.md §8f rank 3); with CPU tensors (the gloo tests of the data-parallel logic) it is `torch.optim.AdamW`, whose
arithmetic the HIP kernels are pinned against (tests/test_optim_gpu.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


def warmup_cosine_lr(step: int, base_lr: float, warmup_steps: int, cosine_steps: int,
                     start_factor: float = 1e-3, eta_min_ratio: float = 0.05) -> float:
    """Closed form of SequentialLR([LinearLR(start_factor, 1, T1), CosineAnnealingLR(T2, eta_min=0.05 lr)], [T1])
    stepped once per iteration (CIFAR100.py:99-113).  Pinned against torch.optim.lr_scheduler in tests/test_host_logic.py."""
    if warmup_steps > 0 and step < warmup_steps:
        return base_lr * (start_factor + (1.0 - start_factor) * step / warmup_steps)
    t = step - warmup_steps
    eta_min = base_lr * eta_min_ratio
    if cosine_steps <= 0:
        return base_lr
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t / cosine_steps)) / 2.0


def cutmix_box(height: int, width: int, lam: float, rng: "np.random.Generator"):
    """Random box covering a (1 - lam) fraction of the image, clipped to it (CutMix; the reference's `utils.rand_bbox`,
    utils.py:1006-1022, which indexes dim 2 then dim 3 of the batch).  Returns (a1, b1, a2, b2): rows a1:a2, cols b1:b2."""
    ratio = math.sqrt(max(0.0, 1.0 - lam))
    cut_a, cut_b = int(height * ratio), int(width * ratio)
    ca, cb = int(rng.integers(height)), int(rng.integers(width))
    a1, a2 = max(ca - cut_a // 2, 0), min(ca + cut_a // 2, height)
    b1, b2 = max(cb - cut_b // 2, 0), min(cb + cut_b // 2, width)
    return a1, b1, a2, b2


@dataclass
class TrainConfig:
    lr: float = 5e-4
    weight_decay: float = 0.05
    grad_max_norm: float = 5.0
    label_smoothing: float = 0.1
    noise_std: float = 0.0
    warmup_steps: int = 0
    cosine_steps: int = 0
    cutmix_prob: float = 0.0       # CIFAR100.py:119-137
    cutmix_beta: float = 1.0
    seed: int = 0


# aten ops whose ROCm implementation goes through thrust / rocPRIM selection primitives (unique-by-key, partition, compaction).
# Round 3's MAE step faulted in the first replay of its captured graph inside rocprim::detail::partition_kernel, launched by
# thrust::unique_by_key_copy from embedding_dense_backward (ROCm debug agent: all resident waves of that kernel with MEM_VIOL;
# bisected over the repository's history to the graph's stream topology, not to a kernel or index of this repository --
# profiles/r04_mae_graph_replay_fault_root_cause.txt).  The step of this repository's models contains none of them (mae.py routes
# its token bookkeeping through the bounds-checked row kernels); a user-supplied compute_loss that does is refused.
_CAPTURE_UNSAFE_OPS = ("aten.embedding_dense_backward", "aten._embedding_bag_dense_backward", "aten.nonzero", "aten.masked_select",
                       "aten._unique", "aten.unique_dim", "aten.unique_consecutive", "aten._unique2", "aten.bincount",
                       "aten.index_put_accumulate")


def _record_unsafe_ops(fn) -> set:
    """Run fn() once (eagerly) and return the names of the ops of _CAPTURE_UNSAFE_OPS it dispatched, forward and backward."""
    from torch.utils._python_dispatch import TorchDispatchMode
    seen = set()

    class _Rec(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if name.startswith(("aten.index_put", "aten._index_put_impl")):
                acc = (kwargs or {}).get("accumulate", args[3] if len(args) > 3 else False)
                if acc:
                    seen.add("aten.index_put(accumulate=True)")
            elif name.startswith(_CAPTURE_UNSAFE_OPS):
                seen.add(name.rsplit(".", 1)[0] if name.count(".") > 1 else name)
            return func(*args, **(kwargs or {}))

    with torch.autograd.set_multithreading_enabled(False), _Rec():      # the backward runs on this thread: the recorder sees it
        fn()
    return seen


class Trainer:
    """One process per GPU.  `reducer` (parallel.GradReducer) is None for single-GPU runs."""

    def __init__(self, model: torch.nn.Module, cfg: TrainConfig, reducer=None, compute_loss=None) -> None:
        """`compute_loss(model, x, y) -> scalar` replaces the classification loss (e.g. `lambda m, x, y: m(x)` for the MAE
        wrapper, whose forward returns its reconstruction loss, mae.py:117-118)."""
        self.model, self.cfg = model, cfg
        self.compute_loss = compute_loss
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        on_gpu = all(p.is_cuda for p in params)
        if on_gpu and reducer is None:
            # single GPU: the reducer is still what keeps all gradients in one flat buffer (it launches no collective)
            from .parallel import GradReducer
            reducer = GradReducer(model, 1)
        self.reducer = reducer
        if on_gpu:
            from .optim import FusedAdamW
            self.opt = FusedAdamW(reducer, lr=cfg.lr, weight_decay=cfg.weight_decay, betas=(0.9, 0.999), eps=1e-8)
        else:
            self.opt = torch.optim.AdamW(params, lr=cfg.lr, weight_decay=cfg.weight_decay, eps=1e-8, betas=(0.9, 0.999))
        self.fused = on_gpu
        self.step_idx = 0
        self._graph = None
        self.rng = np.random.default_rng(cfg.seed)

    def loss_fn(self, logits: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return F.cross_entropy(logits, y, label_smoothing=self.cfg.label_smoothing)

    def forward_backward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if self.cfg.noise_std > 0.0:
            x = x + torch.randn_like(x) * self.cfg.noise_std          # nowak.py:153
        if self.reducer is not None:
            self.reducer.begin_step()
        c = self.cfg
        if self.compute_loss is not None:
            loss = self.compute_loss(self.model, x, y)
        elif c.cutmix_prob > 0.0 and self.rng.random() < c.cutmix_prob:
            # CutMix as in the reference harness (CIFAR100.py:119-137): paste a box from a permuted batch, mix the losses
            perm = torch.randperm(x.shape[0], device=x.device)
            lam = float(self.rng.beta(c.cutmix_beta, c.cutmix_beta))
            a1, b1, a2, b2 = cutmix_box(x.shape[2], x.shape[3], lam, self.rng)
            x = x.clone()
            x[:, :, a1:a2, b1:b2] = x[perm, :, a1:a2, b1:b2]
            lam = 1.0 - (a2 - a1) * (b2 - b1) / float(x.shape[-1] * x.shape[-2])      # exact pixel ratio
            logits = self.model(x)
            loss = self.loss_fn(logits, y) * lam + self.loss_fn(logits, y[perm]) * (1.0 - lam)
        else:
            logits = self.model(x)
            loss = self.loss_fn(logits, y)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish_step()          # waits for the overlapped all-reduces; grads are now rank-averaged
        return loss.detach()

    def optimizer_step(self) -> None:
        c = self.cfg
        lr = warmup_cosine_lr(self.step_idx, c.lr, c.warmup_steps, c.cosine_steps) if (c.warmup_steps or c.cosine_steps) else c.lr
        if self.fused:
            self.opt.step(max_norm=c.grad_max_norm or 0.0, lr=lr)
        else:
            if c.grad_max_norm and c.grad_max_norm > 0:
                torch.nn.utils.clip_grad_norm_(self.params, c.grad_max_norm, foreach=True)
            for g in self.opt.param_groups:
                g["lr"] = lr
            self.opt.step()
            # do not rely on the optimizer bumping the parameters' version counters (torch's fused multi-tensor AdamW on
            # ROCm was seen not to: the bf16 weight images then went stale): drop them explicitly
            from .encoder import WEIGHTS
            WEIGHTS.clear()
            if self.reducer is None:
                self.opt.zero_grad(set_to_none=True)
        self.step_idx += 1

    def step(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if self._graph is not None:
            return self._replay(x, y)
        loss = self.forward_backward(x, y)
        self.optimizer_step()
        return loss

    # ---- whole-step HIP graph -------------------------------------------------------------------
    def capture(self, x: torch.Tensor, y: torch.Tensor, warmup: int = 2, keep_graph: bool = False) -> None:
        """Capture forward + loss + backward + clip + AdamW + weight re-staging (one training step, ~370 - 830 kernel launches)
        into ONE HIP graph; afterwards `step()` copies the batch into the captured input buffers, refreshes three device
        scalars (learning rate, bias corrections) and replays the graph.  What this removes is the host: 13 ms of ctypes /
        autograd enqueue per ViT-S step and the idle gap at the step boundary; the kernels are the same launches.
        Training state is untouched by the capture (the warm-up steps it needs are rolled back), so a captured trainer
        continues exactly where the eager one stood.  Single process only (no collectives are captured), no CutMix (its
        box is drawn on the host every step)."""
        if not self.fused:
            raise RuntimeError("Trainer.capture needs the MI355X path (FusedAdamW)")
        if self.cfg.cutmix_prob > 0.0:
            raise RuntimeError("Trainer.capture: CutMix draws its box on the host every step and cannot be replayed")
        if self.reducer is not None and (self.reducer.world > 1 or self.reducer.force):
            raise RuntimeError("Trainer.capture: collectives are not captured; use the eager step with a process group")
        opt = self.opt
        saved = (opt.param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count, self.step_idx)
        rng_state = torch.cuda.get_rng_state(x.device)
        self._gx, self._gy = x.clone(), y.clone()

        def body():
            loss = self.forward_backward(self._gx, self._gy)
            opt.step(max_norm=self.cfg.grad_max_norm or 0.0, from_device=True)
            return loss

        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        graph = None
        try:
            with torch.cuda.stream(side):
                for it in range(max(warmup, 1)):      # allocator pools, weight-cache job tables, lazy kernel attributes
                    opt.stage_step_scalars(self.cfg.lr)
                    if it > 0:
                        body()
                        continue
                    # the first warm-up step runs under a dispatch recorder: PyTorch library ops that are not safe to replay from a
                    # HIP graph on this stack are refused by NAME, before anything is captured (round 3: the sort-based backward of
                    # nn.Embedding -- thrust::unique_by_key_copy -> rocprim partition_kernel -- faulted in the first replay)
                    seen = _record_unsafe_ops(body)
                    if seen:
                        raise RuntimeError("Trainer.capture: the step runs PyTorch ops that are not replay-safe from a HIP graph on "
                                           f"ROCm ({', '.join(sorted(seen))}); keep such index bookkeeping out of the step (see mae.py) "
                                           "or train this model with the eager step")
            cur.wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()   # kept: tools/graph_nodes.py walks the nodes
            opt.stage_step_scalars(self.cfg.lr)
            with torch.cuda.graph(graph):
                self._gloss = body()
        finally:
            # roll the state back, also when the step was refused: nothing above was a training step
            cur.wait_stream(side)
            with torch.no_grad():
                opt.param.copy_(saved[0]); opt.exp_avg.copy_(saved[1]); opt.exp_avg_sq.copy_(saved[2])
            opt.step_count, self.step_idx = saved[3], saved[4]
            torch.cuda.set_rng_state(rng_state, x.device)
            from .encoder import WEIGHTS
            WEIGHTS.refresh_all()
        self._graph = graph

    def _replay(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        if x.shape != self._gx.shape or y.shape != self._gy.shape:
            # a short last batch (or a batch-1 probe, which copy_ would silently broadcast): the graph holds the captured shapes only
            loss = self.forward_backward(x, y)
            self.optimizer_step()
            return loss
        if x.data_ptr() != self._gx.data_ptr():
            self._gx.copy_(x, non_blocking=True)
        if y.data_ptr() != self._gy.data_ptr():
            self._gy.copy_(y, non_blocking=True)
        lr = warmup_cosine_lr(self.step_idx, c.lr, c.warmup_steps, c.cosine_steps) if (c.warmup_steps or c.cosine_steps) else c.lr
        self.opt.stage_step_scalars(lr)
        self._graph.replay()
        self.step_idx += 1
        return self._gloss.clone()              # the static tensor is overwritten by the next replay: callers may keep losses

    @torch.no_grad()
    def eval_step(self, x: torch.Tensor, y: torch.Tensor, process_group=None) -> torch.Tensor:
        """One evaluation batch as the reference harness does it (examples/CIFAR100.py:148-163): top-1 accuracy of this
        rank's shard, summed onto rank 0 with ONE scalar `reduce` (the C2 collective of SURVEY.md §8e).  Returns the
        accuracy tensor after the reduce: on rank 0 the SUM over ranks (the reference divides by world_size when it logs,
        `after_eval_epoch`), elsewhere this rank's own value."""
        import torch.distributed as dist
        was_training = self.model.training
        self.model.eval()
        try:
            preds = self.model(x)
        finally:
            self.model.train(was_training)
        accu = preds.argmax(1).eq(y).float().mean()
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            dist.reduce(accu, dst=0, group=process_group)
        return accu

    def evaluate(self, batches, process_group=None) -> float:
        """Mean top-1 accuracy over `batches` of (x, y) and over ranks -- meaningful on rank 0 (CIFAR100.py:142-163)."""
        import torch.distributed as dist
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        total, count = 0.0, 0
        for x, y in batches:
            total += self.eval_step(x, y, process_group).item()
            count += 1
        return total / max(count, 1) / world
