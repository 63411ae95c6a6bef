"""Oracle: CPU restatement of the reference ``SimpleViT`` forward (test infrastructure only).

Follows ``/root/reference/vit_pytorch_robust/simple_vit.py``:
  posemb_sincos_2d   :15-28        FeedForward  :34-45
  Attention          :48-76        Transformer  :79-97
  SimpleViT          :100-149
and ``utils.py:1025-1037`` (SinkhornAttention).

Everything is functional: weights come in as a ``state_dict`` with the
reference's key names, so the oracle shares no code with the
product modules.  Pinned by ``tests/golden/simplevit_*.npz`` (generated from the
imported reference by ``tests/golden/gen_golden.py``).

``emulate_bf16=True`` rounds tensors to bf16 at exactly the points where the
HIP path stores/feeds bf16 (DESIGN.md "Numerics"); the arithmetic order is
otherwise unchanged.  It is used to separate "kernel is wrong" from "bf16
operand rounding" in the parity tests; the un-emulated fp32 result remains
the reference value.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _bf16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _Q:
    """Rounding policy: identity (fp32 oracle), bf16 round-trip at every rounding point of the HIP path (`True`), or at a chosen
    subset of them (a collection of point names: "img", "w", "xn", "qkv", "p", "o", "h" -- tools/parity_attribution.py uses
    this to measure which rounding point carries the deviation from the fp32 reference)."""

    POINTS = ("img", "w", "xn", "qkv", "p", "o", "h")

    def __init__(self, emulate):
        self.points = frozenset(self.POINTS) if emulate is True else frozenset(emulate or ())
        unknown = self.points - frozenset(self.POINTS)
        if unknown:
            raise ValueError(f"unknown rounding points {sorted(unknown)}")
        self.emulate = bool(self.points)

    def on(self, point: str) -> bool:
        return point in self.points

    def __call__(self, t: Tensor, point: str = None) -> Tensor:
        if point is None:
            return _bf16(t) if self.emulate else t
        return _bf16(t) if point in self.points else t


def posemb_sincos_2d(h: int, w: int, dim: int, temperature: float = 10000.0) -> Tensor:
    """simple_vit.py:15-28.  Returns [h*w, dim] fp32; token index = row*w + col."""
    assert dim % 4 == 0, "feature dimension must be multiple of 4 for sincos emb"
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    omega = torch.arange(dim // 4) / (dim // 4 - 1)
    omega = 1.0 / (temperature ** omega)
    y = y.flatten()[:, None] * omega[None, :]
    x = x.flatten()[:, None] * omega[None, :]
    pe = torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1)
    return pe.to(torch.float32)


def patchify_p1p2c(img: Tensor, ph: int, pw: int) -> Tensor:
    """einops 'b c (h p1) (w p2) -> b h w (p1 p2 c)' (simple_vit.py:126-129), flattened to [b, h*w, p1*p2*c]."""
    b, c, H, W = img.shape
    h, w = H // ph, W // pw
    t = img.reshape(b, c, h, ph, w, pw)          # b c h p1 w p2
    t = t.permute(0, 2, 4, 3, 5, 1)              # b h w p1 p2 c
    return t.reshape(b, h * w, ph * pw * c)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm over the last dim: biased variance, affine."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (simple_vit.py:40)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def sinkhorn_normalise(P: Tensor, iterations: int = 3) -> Tensor:
    """utils.py:1031-1037 after the softmax: 3x (row /, col /) then a final row /."""
    for _ in range(iterations):
        P = P / P.sum(dim=-1, keepdim=True)
        P = P / P.sum(dim=-2, keepdim=True)
    P = P / P.sum(dim=-1, keepdim=True)
    return P


def attend(q: Tensor, k: Tensor, v: Tensor, scale: float, robust: bool, Q: _Q) -> Tensor:
    """q,k,v [B,H,N,dh] -> [B,H,N,dh].  simple_vit.py:70-74."""
    dots = torch.matmul(q, k.transpose(-1, -2)) * scale
    if not Q.emulate:
        attn = torch.softmax(dots, dim=-1)
        if robust:
            attn = sinkhorn_normalise(attn)
        return torch.matmul(attn, v)
    # bf16 emulation of the fused kernel: unnormalised exp in fp32, P fed to the
    # second product in bf16, normalisation by the fp32 row sum afterwards.
    m = dots.max(dim=-1, keepdim=True).values
    p = torch.exp(dots - m)
    l = p.sum(dim=-1, keepdim=True)
    if robust:
        attn = sinkhorn_normalise(p / l)
        return torch.matmul(_bf16(attn), v)
    return torch.matmul(_bf16(p), v) / l


def attention_block(x: Tensor, sd: Dict[str, Tensor], pfx: str, heads: int, dim_head: int,
                    robust: bool, Q: _Q, eps: float = 1e-5) -> Tensor:
    """Attention.forward, simple_vit.py:64-76 (pre-LN inside; bias-free to_qkv / to_out)."""
    B, N, _ = x.shape
    xn = Q(layer_norm(x, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], eps))
    qkv = Q(xn @ Q(sd[pfx + "to_qkv.weight"]).t())
    q, k, v = qkv.chunk(3, dim=-1)
    # 'b n (h d) -> b h n d'
    q, k, v = (t.reshape(B, N, heads, dim_head).permute(0, 2, 1, 3) for t in (q, k, v))
    out = attend(q, k, v, dim_head ** -0.5, robust, Q)
    out = Q(out.permute(0, 2, 1, 3).reshape(B, N, heads * dim_head))   # 'b h n d -> b n (h d)'
    return out @ Q(sd[pfx + "to_out.weight"]).t()


def feed_forward_block(x: Tensor, sd: Dict[str, Tensor], pfx: str, Q: _Q, eps: float = 1e-5) -> Tensor:
    """FeedForward.forward, simple_vit.py:36-45: LN -> Linear -> GELU -> Linear."""
    xn = Q(layer_norm(x, sd[pfx + "net.0.weight"], sd[pfx + "net.0.bias"], eps))
    u = xn @ Q(sd[pfx + "net.1.weight"]).t() + sd[pfx + "net.1.bias"]
    h = Q(gelu_erf(u))
    return h @ Q(sd[pfx + "net.3.weight"]).t() + sd[pfx + "net.3.bias"]


def depth_of(sd: Dict[str, Tensor]) -> int:
    d = 0
    while f"transformer.layers.{d}.0.norm.weight" in sd:
        d += 1
    return d


def transformer_forward(x: Tensor, sd: Dict[str, Tensor], *, heads: int, dim_head: int = 64,
                        robust: bool = False, emulate_bf16: bool = False,
                        prefix: str = "transformer.", capture: Optional[dict] = None) -> Tensor:
    """Transformer.forward, simple_vit.py:93-97: x = attn(x)+x; x = ff(x)+x, no final norm."""
    Q = _Q(emulate_bf16)
    i = 0
    while f"{prefix}layers.{i}.0.norm.weight" in sd:
        a = attention_block(x, sd, f"{prefix}layers.{i}.0.", heads, dim_head, robust, Q)
        x = a + x
        f = feed_forward_block(x, sd, f"{prefix}layers.{i}.1.", Q)
        x = f + x
        if capture is not None:
            capture[f"layer{i}.attn_branch"] = a.detach().clone()   # Attention.forward output
            capture[f"layer{i}.ff_branch"] = f.detach().clone()     # FeedForward.forward output
            capture[f"layer{i}.out"] = x.detach().clone()           # residual stream after the block
        i += 1
    return x


def simple_vit_forward(sd: Dict[str, Tensor], img: Tensor, *, patch_size: int, heads: int,
                       dim_head: int = 64, robust: bool = False, emulate_bf16: bool = False,
                       capture: Optional[dict] = None) -> Tensor:
    """SimpleViT.forward, simple_vit.py:138-149.  img [B,C,H,W] fp32 -> logits [B,classes]."""
    Q = _Q(emulate_bf16)
    B, C, H, W = img.shape
    p = patch_size
    h, w = H // p, W // p
    wp, bp = sd["to_patch_embedding.1.weight"], sd["to_patch_embedding.1.bias"]
    dim = wp.shape[0]
    patches = patchify_p1p2c(Q(img), p, p)
    x = patches @ Q(wp).t() + bp
    x = x + posemb_sincos_2d(h, w, dim).to(x.dtype)      # simple_vit.py:28: pe.type(dtype)
    if capture is not None:
        capture["embed"] = x.clone()
    x = transformer_forward(x, sd, heads=heads, dim_head=dim_head, robust=robust,
                            emulate_bf16=emulate_bf16, capture=capture)
    x = x.mean(dim=1)
    x = layer_norm(x, sd["linear_head.0.weight"], sd["linear_head.0.bias"], 1e-5)
    return x @ sd["linear_head.1.weight"].t() + sd["linear_head.1.bias"]


def cross_entropy_ls(logits: Tensor, y: Tensor, label_smoothing: float = 0.1) -> Tensor:
    """F.cross_entropy(..., label_smoothing=0.1) restated (examples/CIFAR100.py:139, baseline.py:70)."""
    logp = torch.log_softmax(logits, dim=-1)
    nll = -logp.gather(1, y[:, None]).squeeze(1)
    smooth = -logp.mean(dim=-1)
    return ((1.0 - label_smoothing) * nll + label_smoothing * smooth).mean()


def simple_vit_loss_and_grads(sd: Dict[str, Tensor], img: Tensor, y: Tensor, **kw):
    """Forward + CE(label_smoothing=0.1) + autograd backward on leaf copies of ``sd``.

    Returns (logits, loss, grads-by-key).  Autograd through the restated forward is
    the gradient oracle (the reference has no hand-written backward either).
    """
    leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    logits = simple_vit_forward(leaves, img, **kw)
    loss = cross_entropy_ls(logits, y)
    loss.backward()
    grads = {k: v.grad.detach() for k, v in leaves.items() if v.grad is not None}
    return logits.detach(), loss.detach(), grads
