"""Drop-in `SimpleViT` family running on the MI355X HIP hot path.

Same constructor signatures, forward signatures and state_dict keys as the reference modules
(`/root/reference/vit_pytorch_robust/simple_vit.py`: FeedForward :34-45, Attention :48-76, Transformer :79-97,
SimpleViT :100-149), so a reference checkpoint loads with `load_state_dict` and a training script only changes
its import.  The nn.Linear / nn.LayerNorm children are parameter holders (they give the reference's parameter
names, shapes and initialisers); their own `forward` is never used -- the arithmetic is in libnrv_hip.so via
`encoder.py`.  Modules must live on the HIP device; there is no CPU execution path.
"""
from __future__ import annotations

import torch
from torch import nn

from .encoder import AttnHalfFn, BlockMeta, EncoderStackFn, MlpHalfFn, PatchEmbedFn
from ._lib import PATCH_P1P2C, NrvError


def _pair(t):
    return t if isinstance(t, tuple) else (t, t)


def sincos_table_2d(h: int, w: int, dim: int, temperature: float = 10000.0, device=None) -> torch.Tensor:
    """Fixed 2-D sin/cos positional table [h*w, dim] (simple_vit.py:15-28): [sin(x w), cos(x w), sin(y w), cos(y w)].

    A constant of the model geometry: computed once and folded into the patch-embed GEMM epilogue instead of
    being rebuilt and added on every forward as the reference does (simple_vit.py:142-143).
    """
    if dim % 4:
        raise ValueError("feature dimension must be multiple of 4 for sincos emb")
    quarter = dim // 4
    freq = 1.0 / (temperature ** (torch.arange(quarter, device=device) / (quarter - 1)))
    rows = torch.arange(h, device=device).repeat_interleave(w)      # y of token r*w + c
    cols = torch.arange(w, device=device).repeat(h)                 # x of token r*w + c
    ay = rows[:, None] * freq[None, :]
    ax = cols[:, None] * freq[None, :]
    return torch.cat((ax.sin(), ax.cos(), ay.sin(), ay.cos()), dim=1).to(torch.float32).contiguous()


class _SinkhornNormFn(torch.autograd.Function):
    """softmax + Sinkhorn normalisations of a materialised score tensor through nrv_sinkhorn_fwd / _bwd."""

    @staticmethod
    def forward(ctx, scores, iters: int):
        from . import kernels as K
        if not scores.is_cuda:
            raise NrvError("noise_robust_vit_amd runs on the MI355X (HIP) device only; there is no CPU fallback")
        s32 = scores.detach().to(torch.float32).contiguous()
        out, lse, avec, bvec = K.sinkhorn_fwd(s32, iters)
        ctx.save_for_backward(s32, lse, avec, bvec)
        ctx.iters, ctx.dtype = iters, scores.dtype
        return out.to(scores.dtype)

    @staticmethod
    def backward(ctx, dout):
        from . import kernels as K
        s32, lse, avec, bvec = ctx.saved_tensors
        ds = K.sinkhorn_bwd(s32, dout.to(torch.float32).contiguous(), lse, avec, bvec, ctx.iters)
        return ds.to(ctx.dtype), None


class SinkhornAttention(nn.Module):
    """`robust=True`: softmax followed by `sinkhorn_iterations` x (row, column) normalisations and a final row normalisation
    (utils.py:1025-1037).  Inside the attention modules the arithmetic is fused into the HIP attention kernel and this object
    is only the marker the reference's `Attention.attend` slot holds; called directly on a score tensor [..., R, C] it runs
    the stand-alone kernels of csrc/nrv_sinknorm.hip (forward and backward)."""

    def __init__(self, dim: int = -1, sinkhorn_iterations: int = 3) -> None:
        super().__init__()
        self.dim = dim
        self.sinkhorn_iterations = sinkhorn_iterations

    def forward(self, scores):
        if self.dim not in (-1, scores.dim() - 1):
            raise NotImplementedError("SinkhornAttention: the softmax dimension must be the last one (the reference's default)")
        return _SinkhornNormFn.apply(scores, int(self.sinkhorn_iterations))


class PatchUnfold(nn.Module):
    """Index-0 child of `to_patch_embedding` (parameter-free, keeps the reference key `to_patch_embedding.1.*`)."""

    def __init__(self, patch_height: int, patch_width: int) -> None:
        super().__init__()
        self.patch_height, self.patch_width = patch_height, patch_width

    def forward(self, img):
        from . import kernels as K
        if self.patch_height != self.patch_width:
            raise NotImplementedError("square patches only")
        b, c, hh, ww = img.shape
        p = self.patch_height
        return K.patch_unfold(img, p, PATCH_P1P2C).reshape(b, hh // p, ww // p, c * p * p)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, dim))
        self._meta = BlockMeta(heads=1, dim_head=64, eps=self.net[0].eps)

    def layer_params(self):
        n = self.net
        return [n[0].weight, n[0].bias, n[1].weight, n[1].bias, n[3].weight, n[3].bias]

    def forward(self, x):
        return MlpHalfFn.apply(x, self._meta, *self.layer_params())


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, robust=False):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.attend = SinkhornAttention(-1) if robust else nn.Softmax(dim=-1)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self._meta = BlockMeta(heads=heads, dim_head=dim_head, eps=self.norm.eps, robust=bool(robust))

    def layer_params(self):
        return [self.norm.weight, self.norm.bias, self.to_qkv.weight, None, self.to_out.weight, None]

    def forward(self, x):
        return AttnHalfFn.apply(x, self._meta, *self.layer_params())


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, robust=False):
        super().__init__()
        self.layers = nn.ModuleList(
            nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head, robust=robust), FeedForward(dim, mlp_dim)])
            for _ in range(depth))
        self._meta = BlockMeta(heads=heads, dim_head=dim_head, eps=1e-5, robust=bool(robust))

    def attach_grad_sink(self, sink) -> None:
        self._meta.sink = sink

    def forward(self, x):
        flat = []
        for attn, ff in self.layers:
            flat += attn.layer_params() + ff.layer_params()
        return EncoderStackFn.apply(x, self._meta, *flat)


class SimpleViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim,
                 channels=3, dim_head=64, robust=False):
        super().__init__()
        image_height, image_width = _pair(image_size)
        patch_height, patch_width = _pair(patch_size)
        assert image_height % patch_height == 0 and image_width % patch_width == 0, \
            "Image dimensions must be divisible by the patch size."
        if patch_height != patch_width:
            raise NotImplementedError("the HIP patch-embed path supports square patches")
        self.grid = (image_height // patch_height, image_width // patch_width)
        self.patch = patch_height
        patch_dim = channels * patch_height * patch_width
        self.to_patch_embedding = nn.Sequential(PatchUnfold(patch_height, patch_width), nn.Linear(patch_dim, dim))
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, robust)
        self.to_latent = nn.Identity()
        self.linear_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
        self._pos = {}
        self._sink = None

    def attach_grad_sink(self, sink) -> None:
        """Data-parallel runtime hook: encoder weight gradients are written straight into the sink's buckets."""
        self._sink = sink
        self.transformer.attach_grad_sink(sink)

    def positional_table(self, device, grid=None) -> torch.Tensor:
        """sincos table of the patch grid `grid` (default: the constructor's), built once per (grid, device).  The
        reference recomputes it from the ACTUAL patch grid on every forward (simple_vit.py:141-143), so any image size
        divisible by the patch works there; here each grid seen gets its own cached constant."""
        grid = tuple(self.grid if grid is None else grid)
        key = (grid, str(device))
        tab = self._pos.get(key)
        if tab is None:
            dim = self.to_patch_embedding[1].weight.shape[0]
            tab = self._pos[key] = sincos_table_2d(grid[0], grid[1], dim, device=device)
        return tab

    def forward(self, img):
        lin = self.to_patch_embedding[1]
        if img.dim() != 4 or img.shape[2] % self.patch or img.shape[3] % self.patch:
            raise ValueError(f"expected [B, C, H, W] with H and W divisible by the patch size {self.patch}, got {tuple(img.shape)}")
        grid = (img.shape[2] // self.patch, img.shape[3] // self.patch)
        x = PatchEmbedFn.apply(img, lin.weight, lin.bias, self.positional_table(img.device, grid), None,
                               self.patch, PATCH_P1P2C, self._sink)
        x = self.transformer(x)
        x = x.mean(dim=1)                      # pooling + head stay in PyTorch-ROCm (0.004 % of FLOPs)
        x = self.to_latent(x)
        return self.linear_head(x)
