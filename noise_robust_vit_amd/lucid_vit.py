"""lucidrains-style `ViT` (class token, learned positions, `to_q` / `to_kv` split, biased `to_out`) on the HIP path.

This is the encoder family the reference's `MAE` wrapper expects (`mae.py:29-31`: `to_patch_embedding[:2]`,
`pos_embedding [1, n+1, d]`, `transformer`); module tree, constructor arguments and state_dict keys follow
`/root/reference/vit_pytorch_robust/learnable_memory_vit.py:30-151` (FeedForward keys `net.0,1,4`; Attention keys
`norm, to_q, to_kv, to_out.0`; `Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)`; `ViT(...)`).
Dropout arguments are accepted, p = 0 is what the fused path implements.  The adapter / memory arguments of the
reference's `forward(x, attn_mask, memories)` are outside the hot path and must be None.
"""
from __future__ import annotations

import torch
from torch import nn

from ._lib import PATCH_P1P2C
from .encoder import WEIGHTS, AttnHalfFn, BlockMeta, EncoderStackFn, MlpHalfFn, PatchEmbedFn
from .simple_vit import PatchUnfold, _pair


class _FusedRowsFn(torch.autograd.Function):
    """[wq; wkv] as a view of the owner's persistent fused buffer; backward splits the rows."""

    @staticmethod
    def forward(ctx, owner, wq, wkv):
        ctx.rows = wq.shape[0]
        buf = owner._refresh_fused()
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        return None, g[:ctx.rows], g[ctx.rows:]


class PatchUnfoldFlat(PatchUnfold):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (learnable_memory_vit.py:120)."""

    def forward(self, img):
        x = super().forward(img)
        return x.reshape(x.shape[0], -1, x.shape[-1])


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))
        self.p = dropout
        self._meta = BlockMeta(heads=1, dim_head=64, eps=self.net[0].eps)

    def layer_params(self):
        n = self.net
        return [n[0].weight, n[0].bias, n[1].weight, n[1].bias, n[4].weight, n[4].bias]

    def forward(self, x):
        # stand-alone use (inside a Transformer the stack runs): dropout behind the GELU and behind the second Linear (learnable_memory_vit.py:37,39)
        self._meta.dropout = self.p if self.training else 0.0
        return MlpHalfFn.apply(x, self._meta, *self.layer_params())


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head, self.scale, self.p = heads, dim_head, dim_head ** -0.5, dropout
        self.norm = nn.LayerNorm(dim)
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
        self._meta = BlockMeta(heads=heads, dim_head=dim_head, eps=self.norm.eps)
        self._fused = None           # persistent fp32 [3*inner, dim] copy of [to_q.weight; to_kv.weight]
        self._fused_key = None
        WEIGHTS.add_derived(self._refresh_fused)

    def _refresh_fused(self):
        """The fused projection [to_q.weight; to_kv.weight] as ONE fp32 tensor.
        * optim.FusedAdamW keeps the parameters in a flat buffer with the gradient slots' layout, and grad_groups() puts to_q /
          to_kv back to back there: the fused projection is then simply a VIEW of that memory -- no copy, never stale;
        * otherwise a persistent buffer, rebuilt (two row-block copies, in place) only when a source changed -- version
          counters / addresses, or the weight cache's epoch for updates through raw pointers."""
        wq, wkv = self.to_q.weight, self.to_kv.weight
        adjacent = (wq.is_contiguous() and wkv.is_contiguous() and wq.dtype == torch.float32 and wkv.dtype == torch.float32
                    and wq.untyped_storage().data_ptr() == wkv.untyped_storage().data_ptr()
                    and wkv.data_ptr() == wq.data_ptr() + wq.numel() * 4)
        if adjacent:
            # the versions are part of the key: an in-place write to the parameters (load_state_dict, p.copy_()) bumps THEIR
            # counters, not the counter of a separate tensor over the same storage, and the weight cache is keyed on
            # (owner, version) -- a fresh tensor object makes it miss and re-stage the bf16 images.  (Updates through raw
            # pointers by optim.FusedAdamW leave the versions alone and are followed by WEIGHTS.refresh_all().)
            key = ("view", wq.data_ptr(), wkv.data_ptr(), wq._version, wkv._version)
            if self._fused_key != key:
                # a tensor of its own over the same storage (not a view with a `_base`: the weight cache tracks owners)
                self._fused = torch.empty(0, dtype=torch.float32, device=wq.device).set_(
                    wq.untyped_storage(), wq.storage_offset(), (wq.shape[0] + wkv.shape[0], wq.shape[1]), (wq.shape[1], 1))
                self._fused_key = key
            return self._fused
        key = (wq._version, wkv._version, wq.data_ptr(), wkv.data_ptr(), WEIGHTS.epoch)
        if self._fused is None or self._fused.device != wq.device or (isinstance(self._fused_key, tuple) and self._fused_key[:1] == ("view",)):
            self._fused = torch.empty(wq.shape[0] + wkv.shape[0], wq.shape[1], dtype=torch.float32, device=wq.device)
            self._fused_key = None
        if self._fused_key != key:
            with torch.no_grad():
                self._fused[:wq.shape[0]].copy_(wq)
                self._fused[wq.shape[0]:].copy_(wkv)
            self._fused_key = key
        return self._fused

    def layer_params(self, for_sink: bool = False):
        # one [3*inner, dim] projection for the fused QKV GEMM: a persistent buffer whose bf16 images stay cached across
        # forwards (a fresh torch.cat per call would be re-cast and re-transposed every time).  Under autograd _FusedRowsFn
        # hands the gradient rows back to to_q / to_kv; with a gradient sink (parallel.GradReducer) the buffer itself is
        # passed, tagged with the parameters whose rows it stacks: the backward writes each block into its slot directly
        if for_sink:
            wqkv = self._refresh_fused()
            rows = self.to_q.weight.shape[0]
            wqkv._nrv_parts = [(self.to_q.weight, 0, rows), (self.to_kv.weight, rows, wqkv.shape[0])]
        else:
            wqkv = _FusedRowsFn.apply(self, self.to_q.weight, self.to_kv.weight)
        return [self.norm.weight, self.norm.bias, wqkv, None, self.to_out[0].weight, self.to_out[0].bias]

    def forward(self, x, attn_mask=None, memories=None):
        if attn_mask is not None or memories is not None:
            raise NotImplementedError("attention masks / memory tokens are outside the encoder hot path")
        # stand-alone use: dropout on the attention weights (learnable_memory_vit.py:83; composed on the materialised matrix) and behind to_out (:61)
        self._meta.dropout = self._meta.attn_dropout = self.p if self.training else 0.0
        return AttnHalfFn.apply(x, self._meta, *self.layer_params())


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList(
            nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout), FeedForward(dim, mlp_dim, dropout=dropout)])
            for _ in range(depth))
        self.p = dropout
        self._meta = BlockMeta(heads=heads, dim_head=dim_head, eps=1e-5)

    def grad_groups(self):
        """Parameters whose gradients one kernel writes as consecutive row blocks: parallel.GradReducer lays their slots out
        back to back in this order."""
        return [(attn.to_q.weight, attn.to_kv.weight) for attn, _ in self.layers]

    def attach_grad_sink(self, sink) -> None:
        """The layers' weight gradients go straight into the sink's flat buffer (parallel.GradReducer), as in
        simple_vit.Transformer / vit.Encoder: no per-parameter fill + accumulate pass through autograd."""
        self._meta.sink = sink

    def forward(self, x, attn_mask=None, memories=None):
        if attn_mask is not None or memories is not None:
            raise NotImplementedError("attention masks / memory tokens are outside the encoder hot path")
        # the one `dropout` of the reference's Transformer (learnable_memory_vit.py:90-96) sits at four places per layer: on the attention
        # weights (:83, composed on the materialised matrix), behind to_out (:61), behind the GELU (:37) and behind the second Linear (:39)
        self._meta.dropout = self._meta.attn_dropout = self.p if self.training else 0.0
        # a sink only receives gradients of a training step: under no_grad (evaluation) nothing is written anyway
        for_sink = self._meta.sink is not None
        flat = []
        for attn, ff in self.layers:
            flat += attn.layer_params(for_sink) + ff.layer_params()
        return EncoderStackFn.apply(x, self._meta, *flat)


class ViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls', channels=3,
                 dim_head=64, dropout=0., emb_dropout=0.):
        super().__init__()
        ih, iw = _pair(image_size)
        ph, pw = _pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        if ph != pw:
            raise NotImplementedError("square patches only")
        num_patches = (ih // ph) * (iw // pw)
        self.patch = ph
        self.to_patch_embedding = nn.Sequential(PatchUnfoldFlat(ph, pw), nn.Linear(channels * ph * pw, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))

    def grad_groups(self):
        return self.transformer.grad_groups()

    def attach_grad_sink(self, sink) -> None:
        self.transformer.attach_grad_sink(sink)

    def img_to_tokens(self, img):
        lin = self.to_patch_embedding[1]
        x = PatchEmbedFn.apply(img, lin.weight, lin.bias, self.pos_embedding, self.cls_token, self.patch, PATCH_P1P2C, None)
        if self.training and self.dropout.p > 0.0:                  # emb_dropout (learnable_memory_vit.py:147): site -1
            from .vit import _input_dropout
            x = _input_dropout(x, self.transformer._meta, self.dropout.p)
        return x

    def forward(self, img):
        x = self.transformer(self.img_to_tokens(img))
        return self.mlp_head(x[:, 0])
