"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI, overlapped with backward.

The reference trains with DDP inside an external trainer (examples/CIFAR100.py:22-28,171,206-208: global batch
split evenly over 8 ranks, gradients averaged).  This is the MI355X-side counterpart, built for the way the HIP
encoder path produces gradients:

  * all gradients live in ONE flat fp32 buffer; `param.grad` are views into it, so there is no
    bucket-copy pass and the optimizer reads the reduced values in place;
  * the encoder's hand-scheduled backward writes each layer's weight gradients straight into its views
    (`target()`), then calls `layer_done()`; as soon as every parameter of a bucket is final the bucket's
    `all_reduce` is enqueued (async) -- it runs on the process group's RCCL stream, ordered after the producing
    kernels by an event, while the remaining backward GEMMs keep the compute stream busy;
  * parameters handled by PyTorch autograd (classifier head, positional table, class token) are tracked with
    post-accumulate hooks and flushed with the last bucket;
  * buckets follow backward order (head first, patch-embed last) and are sized for xGMI: a few large
    messages (default 64 MiB) rather than NVSwitch-style many small ones -- ring all-reduce on a fully
    connected 8-GPU xGMI mesh is per-link bound, so fewer, larger collectives amortise the launch and
    keep every link streaming;
  * `finish_step()` waits for all collectives (compute stream waits on the RCCL stream) before clip + AdamW;
  * knobs sized for xGMI (7 point-to-point links x ~153 GB/s per GPU, ring collectives per-link bound): `grad_dtype="bf16"`
    halves the bytes on the links (ViT-B/16: 173 instead of 346 MB per step; each bucket's range of ONE persistent bf16 image of the flat buffer is
    written by nrv_cast_f32_bf16 and reduced in place; on the HIP device the optimizer reads that image directly), `tail_mib` caps the LAST
    bucket in backward order (patch embedding / first layers: the only one with no backward work left to hide behind),
    `make_process_group(..., max_ctas=...)` bounds the CUs RCCL's kernels may occupy next to GEMMs that hold every CU.

Works with the `gloo` backend on CPU tensors too (tests/test_parallel_gloo.py): there all parameters are
autograd-managed and averaging is SUM followed by a scale.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


SLOT_ALIGN = 8        # elements: 32 bytes of the fp32 buffer, 16 bytes of its bf16 image (`flat16`) -- kernel operands need 16


def _slot_len(n: int) -> int:
    return (n + SLOT_ALIGN - 1) // SLOT_ALIGN * SLOT_ALIGN


def make_process_group(rank: int, world: int, device=None, backend: str = "nccl", max_ctas: int = 0, min_ctas: int = 0,
                       high_priority_stream: bool = False, **kw):
    """`dist.init_process_group` with the RCCL communicator's CU budget made explicit ("nccl" IS RCCL on ROCm).
    `max_ctas` / `min_ctas` (0: RCCL's default) go into the communicator config (ncclConfig_t maxCTAs / minCTAs = the
    workgroups = channels a collective may launch): the GEMMs of the backward occupy all 256 CUs with one workgroup each, so
    every CU an all-reduce takes is a CU a GEMM round waits for; 8-16 workgroups keep a ring over 7 xGMI links busy.
    Returns the options' description (for bench.py's `config.collectives`)."""
    opts = None
    desc = f"{backend} default communicator config"
    if backend == "nccl" and (max_ctas or min_ctas or high_priority_stream):
        # the fallback is decided BEFORE the rendezvous, from what this torch / RCCL build exposes: errors of
        # init_process_group itself (rendezvous timeout, port in use, device binding) propagate unchanged
        why = None
        if not hasattr(dist, "ProcessGroupNCCL") or not hasattr(dist.ProcessGroupNCCL, "Options"):
            why = "torch.distributed has no ProcessGroupNCCL.Options"
        else:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=bool(high_priority_stream))
            cfg = getattr(opts, "config", None)
            if (max_ctas or min_ctas) and (cfg is None or not hasattr(cfg, "max_ctas") or not hasattr(cfg, "min_ctas")):
                why, opts = "ProcessGroupNCCL.Options has no config.max_ctas / min_ctas", None
        if opts is not None:
            if max_ctas:
                opts.config.max_ctas = int(max_ctas)
            if min_ctas:
                opts.config.min_ctas = int(min_ctas)
            desc = f"rccl max_ctas={max_ctas or 'default'} min_ctas={min_ctas or 'default'} high_priority_stream={bool(high_priority_stream)}"
        else:
            desc = f"{backend} default communicator config (requested max_ctas={max_ctas} min_ctas={min_ctas} not applied: {why})"
    args = dict(rank=rank, world_size=world, **kw)
    if device is not None and backend == "nccl":
        args["device_id"] = device
    if opts is not None:
        args["pg_options"] = opts
    dist.init_process_group(backend, **args)
    return desc


class GradReducer:
    def __init__(self, model: torch.nn.Module, world: int, bucket_mib: float = 64.0,
                 process_group=None, attach: bool = True, sync_params: bool = True,
                 force_collectives: bool = False, grad_dtype: str = "fp32", tail_mib: float = 8.0,
                 reserve_cus: int = 0) -> None:
        """`sync_params`: broadcast rank 0's parameters and buffers at construction, as DistributedDataParallel does
        (examples/CIFAR100.py:206-208 wraps the model in DDP): replicas start identical whatever each rank's seed was.
        `force_collectives`: issue every bucket's all-reduce even when world == 1 (a single-GPU RCCL process group
        exercises exactly the calls, stream ordering and buffer slicing of the multi-GPU step).
        `grad_dtype`: "fp32" (default, exact mean of the ranks' fp32 gradients; `param.grad` holds the reduced values after
        finish_step) or "bf16" (on the HIP device the reduced means live in `grad_buffer()` = the bf16 image `flat16`, which
        optim.FusedAdamW reads in place; `param.grad` then still shows this rank's LOCAL fp32 gradients) -- each rank's bucket is rounded to
        bf16 before the reduction: 8 significant bits per addend, relative error of the mean <= 2^-8 per element -- the size of
        the rounding the bf16 GEMM operands already carry; tests/test_parallel_gloo.py states the measured cost).
        `tail_mib`: upper bound of the last bucket in backward order (0: no cap).
        `reserve_cus`: CUs the GEMM launches leave to the collective's kernels (`nrv_set_reserved_cus`, process-wide; only
        with world > 1 on the HIP device).  The NT GEMM holds one workgroup on every CU it plans for during a whole launch and
        the TN GEMM sizes its splits to one round of them: planned for all 256 with RCCL's kernels resident on some, their
        last workgroups would run in a second round.  Pair it with `make_process_group(max_ctas=...)`."""
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_dtype must be 'fp32' or 'bf16'")
        self.world = world
        self.pg = process_group
        self.force = bool(force_collectives)
        self.grad_dtype = grad_dtype
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        # applied from the first bucket's all-reduce to the end of the step (_launch / finish_step): the forward pass and the
        # backward's first layers, which run beside no collective, keep every CU
        self.reserve_cus = int(reserve_cus) if (world > 1 and params[0].is_cuda) else 0
        self._reserved = False
        if sync_params and world > 1 and dist.is_initialized():
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
            # the broadcast writes through .data (no version bump): bf16 weight images staged by a forward that ran before
            # this point (a sanity / warm-up forward on a rank != 0) would keep the pre-broadcast values
            from .encoder import WEIGHTS
            WEIGHTS.clear()
        # backward order: the reverse of registration order (head ... first encoder layer ... patch embed); parameters the
        # model wants back to back in a given order (`grad_groups()`: lucid_vit's to_q / to_kv, whose gradients are the row
        # blocks of ONE fused-projection GEMM) are placed as a unit where the first of them falls
        order = list(reversed(params))
        groups = [tuple(g) for g in model.grad_groups()] if hasattr(model, "grad_groups") else []
        if groups:
            group_of = {id(q): g for g in groups for q in g}
            placed, regrouped = set(), []
            for q in order:
                if id(q) in placed:
                    continue
                for m in group_of.get(id(q), (q,)):
                    regrouped.append(m)
                    placed.add(id(m))
            order = regrouped
        dev = params[0].device
        sizes = [p.numel() for p in order]
        # slots of SLOT_ALIGN elements: every view (and every bucket range of the bf16 image) can be a kernel operand
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += _slot_len(n)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._slot: Dict[int, Tuple[int, int]] = {}
        self._views: Dict[int, torch.Tensor] = {}
        for p, o, n in zip(order, offs, sizes):
            v = self.flat[o:o + n].view(p.shape)
            self._slot[id(p)] = (o, n)
            self._views[id(p)] = v
            p.grad = v
        # buckets = contiguous ranges in backward order
        cap = int(bucket_mib * (1 << 20) / 4)
        self.buckets: List[Tuple[int, int, List[int]]] = []      # (start, end, [param ids])
        cur_start, cur_ids = 0, []
        for p, o, n in zip(order, offs, sizes):
            cur_ids.append(id(p))
            end = o + _slot_len(n)
            if end - cur_start >= cap:
                self.buckets.append((cur_start, end, cur_ids))
                cur_start, cur_ids = end, []
        if cur_ids:
            self.buckets.append((cur_start, total, cur_ids))
        # The last bucket cannot overlap anything (its parameters are final when the backward ends): split it so that what
        # is left after the split point is at most `tail_mib` -- the bulk goes out while the first layers' backward runs.
        tail_cap = int(tail_mib * (1 << 20) / 4)
        if tail_cap > 0 and self.buckets:
            s0, e0, ids0 = self.buckets[-1]
            if e0 - s0 > tail_cap and len(ids0) > 1:
                k = len(ids0) - 1                          # at least the last parameter (it may be larger than the cap by itself)
                for j in range(1, len(ids0)):
                    if e0 - self._slot[ids0[j]][0] <= tail_cap:
                        k = j
                        break
                cut = self._slot[ids0[k]][0]
                self.buckets[-1] = (s0, cut, ids0[:k])
                self.buckets.append((cut, e0, ids0[k:]))
        self._stage_slabs: list = []
        self.flat16: Optional[torch.Tensor] = None        # bf16 image of `flat` (grad_dtype="bf16": allocated at the first exchange)
        self.reduced_in_bf16 = False
        self._bucket_of = {pid: b for b, (_, _, ids) in enumerate(self.buckets) for pid in ids}
        self._sink_managed: set = set()
        self._ready: set = set()
        self._launched: List[bool] = []
        self._works: list = []
        self._avg_native = dist.is_initialized() and dist.get_backend(self.pg) == "nccl"
        self._params = params
        for p in params:
            p.register_post_accumulate_grad_hook(self._autograd_hook)
        if attach and hasattr(model, "attach_grad_sink"):
            model.attach_grad_sink(self)
        self.begin_step()

    def parameters(self):
        """The trainable parameters, registration order."""
        return list(self._params)

    def params_without_grad(self):
        """Parameters that received no gradient in the step just finished (valid between finish_step and begin_step):
        torch.optim.AdamW skips those entirely (no weight decay, no moment update), and so does optim.FusedAdamW."""
        if self.world > 1:
            return []          # decided identically on all ranks without a collective: see finish_step
        return [p for p in self._params if id(p) not in self._ready]

    def slot(self, p: torch.Tensor):
        """(offset, numel) of p's gradient in `flat` (optim.FusedAdamW lays its parameter / moment buffers out the same way)."""
        return self._slot[id(p)]

    # ---- sink interface used by encoder.py ---------------------------------------------------
    def target(self, p: torch.Tensor):
        """Where a kernel should write d(loss)/d(p) this step, and the beta to use (0: overwrite)."""
        self._sink_managed.add(id(p))
        return self._views[id(p)], 0.0

    def target_block(self, params: Sequence[torch.Tensor]):
        """One [sum of rows, cols] view over the slots of `params` when they lie back to back in this order (grad_groups),
        and the beta to use; None otherwise (the caller then writes each parameter's block separately)."""
        start, _ = self._slot[id(params[0])]
        end = start
        for q in params:
            o, n = self._slot[id(q)]
            if o != end or n % SLOT_ALIGN or q.dim() != 2 or q.shape[1] != params[0].shape[1]:
                return None
            end = o + n
        for q in params:
            self._sink_managed.add(id(q))
        return self.flat[start:end].view(-1, params[0].shape[1]), 0.0

    def layer_done(self, layer_index: int, params: Sequence[torch.Tensor]) -> None:
        for p in params:
            self._mark(id(p))

    # ---- autograd-managed parameters ---------------------------------------------------------
    def _autograd_hook(self, p: torch.Tensor) -> None:
        if id(p) in self._sink_managed:
            return
        if p.grad is not self._views[id(p)]:
            # autograd replaced the view (first accumulation into a None grad): copy back and restore the view
            self._views[id(p)].copy_(p.grad)
            p.grad = self._views[id(p)]
        self._mark(id(p))

    def _mark(self, pid: int) -> None:
        if pid in self._ready:
            return
        self._ready.add(pid)
        b = self._bucket_of[pid]
        if not self._launched[b] and all(q in self._ready for q in self.buckets[b][2]):
            self._launch(b)

    def _launch(self, b: int) -> None:
        self._launched[b] = True
        if (self.world <= 1 and not self.force) or not dist.is_initialized():
            return
        s, e, _ = self.buckets[b]
        if self.reserve_cus > 0 and not self._reserved:
            from . import kernels as K
            K.set_reserved_cus(self.reserve_cus)
            self._reserved = True
        op = dist.ReduceOp.AVG if self._avg_native else dist.ReduceOp.SUM
        if self.grad_dtype == "bf16":
            # the bucket's range of ONE persistent bf16 image of the flat buffer (same slot layout): cast on the compute stream,
            # behind the kernels that produced the gradients, in front of the collective.  On the HIP device the reduced image is
            # what optim.FusedAdamW reads (nrv_sumsq_f32 / nrv_adamw_f32 take bf16 gradients in place, ABI 11): nothing is
            # converted back into the fp32 buffer.
            if self.flat16 is None:
                self.flat16 = torch.zeros(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
            slab = self.flat16[s:e]
            if self.flat.is_cuda:
                from . import kernels as K
                K.cast_bf16(self.flat[s:e], out=slab)
            else:
                slab.copy_(self.flat[s:e])
            self._stage_slabs.append((s, e, slab))
            self._works.append(dist.all_reduce(slab, op=op, group=self.pg, async_op=True))
            return
        self._works.append(dist.all_reduce(self.flat[s:e], op=op, group=self.pg, async_op=True))

    # ---- step protocol -----------------------------------------------------------------------
    def begin_step(self) -> None:
        self.reduced_in_bf16 = False
        if self._reserved:                      # a backward that raised after the first bucket never reached finish_step: the
            from . import kernels as K          # process-wide CU reservation must not outlive its step
            K.set_reserved_cus(0)
            self._reserved = False
        self._ready = set()
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._stage_slabs = []
        # autograd accumulates (+=) into existing .grad views: clear the ones it manages -- as the few contiguous ranges they
        # form in the flat buffer (head ... | ... positions, class token), not one fill kernel per parameter
        for v in self._autograd_ranges():
            v.zero_()
        for p in self._params:
            if id(p) not in self._sink_managed and p.grad is not self._views[id(p)]:
                p.grad = self._views[id(p)]

    def _autograd_ranges(self) -> List[torch.Tensor]:
        """Views over the maximal runs of slots whose gradients autograd accumulates (recomputed when the set of kernel-written
        parameters grew: it is only known after the first backward)."""
        key = len(self._sink_managed)
        if getattr(self, "_zr_key", None) != key:
            spans = sorted((o, o + _slot_len(n)) for pid, (o, n) in self._slot.items() if pid not in self._sink_managed)
            runs: List[List[int]] = []
            for a, b in spans:
                if runs and runs[-1][1] == a:
                    runs[-1][1] = b
                else:
                    runs.append([a, b])
            self._zr = [self.flat[a:b] for a, b in runs]
            self._zr_key = key
        return self._zr

    def finish_step(self) -> None:
        # Parameters that received no gradient on THIS rank this step.  Their slots must not carry anything into the
        # reduction or the clip norm: autograd-managed slots were zeroed in begin_step, kernel-written (sink-managed) slots
        # still hold the previous step's gradient -- zero them now.  With world > 1 another rank may have used the parameter:
        # the reduced gradient (this rank contributing zeros, DDP's find_unused_parameters semantics) is applied on EVERY
        # rank, so replicas cannot diverge (`params_without_grad` is empty then); a parameter unused on every rank thus
        # takes a zero-gradient AdamW step where torch.optim.AdamW would skip it -- the one documented difference.
        for p in self._params:
            if id(p) not in self._ready and id(p) in self._sink_managed:
                self._views[id(p)].zero_()
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for w in self._works:
            w.wait()
        self._works = []
        if self._reserved:
            from . import kernels as K
            K.set_reserved_cus(0)
            self._reserved = False
        exchanged = bool(self._stage_slabs)
        self.reduced_in_bf16 = exchanged and self.flat.is_cuda and self._avg_native
        if exchanged and not self.reduced_in_bf16:
            # CPU tensors (gloo tests; torch.optim.AdamW reads the fp32 views) or a backend without AVG: the reduced bf16 means go
            # back into the fp32 buffer.  On the HIP device with RCCL the optimizer reads `flat16` itself (`grad_buffer()`).
            for s, e, slab in self._stage_slabs:
                self.flat[s:e].copy_(slab)
        self._stage_slabs = []
        if (self.world > 1 or self.force) and not self._avg_native and dist.is_initialized():
            self.flat.mul_(1.0 / self.world)

    def grad_buffer(self) -> torch.Tensor:
        """The flat buffer that holds the step's final gradients (valid between finish_step and begin_step): the fp32 buffer, or
        -- after a bf16 exchange on the HIP device -- its reduced bf16 image, which every bucket has overwritten."""
        return self.flat16 if self.reduced_in_bf16 else self.flat

    def bucket_bytes(self) -> List[int]:
        return [(e - s) * 4 for s, e, _ in self.buckets]
