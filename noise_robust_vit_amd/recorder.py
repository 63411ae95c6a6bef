"""Attention-map recorder with the reference's interface (vit_pytorch_robust/recorder.py:12-60):

    v = Recorder(v)
    preds, attns = v(img)          # attns: [batch, depth, heads, tokens, tokens]
    v = v.eject()                  # back to the plain module

The reference hooks each `Attention.attend` (the Softmax module) and clones its output.  The HIP path never
materialises the [N, N] matrix, so there is nothing to hook: while a Recorder runs the model, the encoder recomputes
each layer's probabilities from q, k and the saved log-sum-exp with `nrv_attn_probs` (and applies the saved Sinkhorn
scalings for robust=True).  Works for every module of this package (SimpleViT, VisionTransformer, lucid ViT, MAE encoder).
"""
from __future__ import annotations

import torch
from torch import nn

from . import encoder


class Recorder(nn.Module):
    def __init__(self, vit: nn.Module, device=None) -> None:
        super().__init__()
        self.vit = vit
        self.data = None
        self.recordings = []
        self.ejected = False
        self.device = device

    def eject(self) -> nn.Module:
        self.ejected = True
        self.recordings.clear()
        return self.vit

    def clear(self) -> None:
        self.recordings.clear()

    def record(self, attn: torch.Tensor) -> None:
        self.recordings.append(attn.clone().detach())

    def forward(self, img: torch.Tensor):
        assert not self.ejected, "recorder has been ejected, cannot be used anymore"
        self.clear()
        with encoder.record_attention(self.recordings):
            pred = self.vit(img)
        target = self.device if self.device is not None else img.device
        recs = tuple(t.detach().to(target) for t in self.recordings)
        attns = torch.stack(recs, dim=1) if len(recs) > 0 else None
        return pred, attns
