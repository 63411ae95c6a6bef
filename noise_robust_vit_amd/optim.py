"""Fused optimizer step of the training harness: global-norm clipping + AdamW on flat fp32 buffers (csrc/nrv_optim.hip).

Counterpart of what the reference harness runs after backward (examples/CIFAR100.py:90-97,191-192; baseline.py:127):
`torch.nn.utils.clip_grad_norm_(params, 5.0)` followed by `torch.optim.AdamW(...).step()`.  The arithmetic is
torch.optim.AdamW's (pinned against it in tests/test_optim_gpu.py); the layout is this build's:

  * gradients already live in ONE flat fp32 buffer (`parallel.GradReducer.flat`, backward order, 16-byte slots);
  * parameters are moved into a second flat buffer with the SAME slot layout (each `param.data` becomes a view, so
    `state_dict()` / checkpoints are unchanged), Adam moments are two more;
  * one step = `nrv_sumsq_f32` over the gradient buffer + `nrv_adamw_f32` over all four: ~32 B/parameter of HBM traffic
    and 3 launches instead of PyTorch's ~6 multi-tensor launches over 150 tensors; the clip coefficient never visits the host.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import kernels as K
from .parallel import _slot_len
from ._lib import NrvError


class FusedAdamW:
    def __init__(self, reducer, lr: float, weight_decay: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        if not reducer.flat.is_cuda:
            raise NrvError("FusedAdamW runs on the MI355X only (flat gradient buffer is not a HIP tensor)")
        self.reducer = reducer
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), (float(betas[0]), float(betas[1])), float(eps)
        self.grad = reducer.flat
        dev = self.grad.device
        self.param = torch.zeros_like(self.grad)
        self._views: Dict[int, torch.Tensor] = {}
        for p in reducer.parameters():
            o, n = reducer.slot(p)
            view = self.param[o:o + n].view(p.shape)
            with torch.no_grad():
                view.copy_(p.data)
            p.data = view                                   # the module now reads / writes the flat buffer
            self._views[id(p)] = view
        self.exp_avg = torch.zeros_like(self.grad)
        self.exp_avg_sq = torch.zeros_like(self.grad)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.empty(max(K.sumsq_workspace(self.grad.numel()) // 4, 4), dtype=torch.float32, device=dev)
        self.step_count = 0
        # step-dependent scalars in device memory (graph replay): { 1 - lr wd, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) }
        self._dyn = torch.zeros(4, dtype=torch.float32, device=dev)

    def stage_step_scalars(self, lr: float = None) -> None:
        """Advance the step count and put this step's learning rate / bias corrections into device memory: what
        `step(..., from_device=True)` -- the launch a captured HIP graph replays -- reads instead of host arguments.
        Same double-precision arithmetic as the C ABI's host side (csrc/nrv_optim.hip)."""
        self.step_count += 1
        lr = self.lr if lr is None else float(lr)
        b1, b2 = self.betas
        t = self.step_count
        # a fresh pinned staging tensor per step: the host runs several replays ahead of the device, and the caching host
        # allocator does not hand a pinned block out again before the asynchronous copy that reads it has run
        host = torch.tensor([1.0 - lr * self.weight_decay, lr / (1.0 - math.pow(b1, t)), 1.0 / math.sqrt(1.0 - math.pow(b2, t)), 0.0],
                            dtype=torch.float32).pin_memory()
        self._dyn.copy_(host, non_blocking=True)

    def _check_layout(self) -> None:
        for p in self.reducer.parameters():
            if p.data_ptr() != self._views[id(p)].data_ptr():
                raise NrvError("a parameter was re-allocated after FusedAdamW was built (e.g. model.to(...)): "
                               "build the optimizer after the model is on its device")

    def step(self, max_norm: float = 0.0, lr: float = None, from_device: bool = False) -> None:
        """One AdamW step on every parameter; max_norm > 0 clips the global gradient norm first (clip_grad_norm_).
        `from_device`: the step count, learning rate and bias corrections come from `stage_step_scalars()` (device memory)
        instead of this call's arguments -- the form train.Trainer captures into a HIP graph."""
        if self.step_count == 0 or (from_device and self.step_count == 1):
            self._check_layout()
        if not from_device:
            self.step_count += 1
        grad = self.reducer.grad_buffer() if hasattr(self.reducer, "grad_buffer") else self.grad     # fp32, or the reduced bf16 image
        if max_norm and max_norm > 0:
            K.sumsq(grad, self.gnorm_sq, self._ws)
        # parameters without a gradient this step are skipped like torch.optim.AdamW skips `p.grad is None` (no weight decay,
        # no moment update): the flat buffers are walked in the ranges between their slots (normally ONE range).  Single
        # process only -- with a process group every parameter steps (parallel.GradReducer.finish_step).  The bias
        # correction uses the optimizer's global step count; torch keeps a per-parameter count, which differs only for a
        # parameter that skipped steps.
        for lo, hi in self._active_ranges():
            K.adamw_flat(self.param[lo:hi], grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                         self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                         max(self.step_count, 1), self.gnorm_sq if max_norm and max_norm > 0 else None, float(max_norm or 0.0),
                         step_scalars=self._dyn if from_device else None)
        # the parameters changed through raw pointers: their version counters did not move, so the bf16 weight images
        # (encoder.WeightCache, keyed on the version) are re-staged explicitly -- all of them in one batched launch
        from .encoder import WEIGHTS
        WEIGHTS.refresh_all()

    def _active_ranges(self):
        skip = sorted(self.reducer.slot(p) for p in self.reducer.params_without_grad())
        if not skip:
            return [(0, self.grad.numel())]
        out, cur = [], 0
        for o, n in skip:
            if o > cur:
                out.append((cur, o))
            cur = o + _slot_len(n)
        if cur < self.grad.numel():
            out.append((cur, self.grad.numel()))
        return out

    def grad_norm(self) -> torch.Tensor:
        """Global gradient norm seen by the last clipped step (device scalar; reading it synchronises)."""
        return self.gnorm_sq.sqrt()

    def _layout(self):
        """[(offset, numel)] of every parameter in registration order: what the flat moment buffers mean."""
        return [tuple(self.reducer.slot(p)) for p in self.reducer.parameters()]

    def state_dict(self) -> dict:
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "layout": self._layout(),
                "lr": self.lr, "weight_decay": self.weight_decay, "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd: dict) -> None:
        """The moments are flat buffers in the reducer's slot order, which depends on `grad_groups()` and on the build that
        wrote the checkpoint: the saved slot table is compared with this optimizer's and the moments are re-laid out per
        parameter when it differs (same parameters in the same registration order, other offsets); anything else raises."""
        mine, theirs = self._layout(), sd.get("layout")
        if theirs is None:
            if sd["exp_avg"].numel() != self.exp_avg.numel():
                raise NrvError("optimizer state without a slot table and of another size: cannot be loaded")
            theirs = mine           # checkpoints written before the table existed: same build, same layout
        theirs = [tuple(t) for t in theirs]
        if len(theirs) != len(mine) or any(a[1] != b[1] for a, b in zip(mine, theirs)):
            raise NrvError("optimizer state belongs to another parameter set (count or sizes differ)")
        self.step_count = int(sd["step"])
        if theirs == mine and sd["exp_avg"].numel() == self.exp_avg.numel():
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            return
        src_m, src_v = sd["exp_avg"].to(self.exp_avg.device), sd["exp_avg_sq"].to(self.exp_avg.device)
        for (o, n), (so, sn) in zip(mine, theirs):
            self.exp_avg[o:o + n].copy_(src_m[so:so + sn])
            self.exp_avg_sq[o:o + n].copy_(src_v[so:so + sn])
