"""Oracle: CPU restatement of the reference torchvision-style ``VisionTransformer`` (test infrastructure only).

Structure follows ``/root/reference/vit_pytorch_robust/vit.py``:
  MLPBlock      :35-84    (Linear, GELU, Dropout, Linear, Dropout; keys ``mlp.0`` / ``mlp.3``)
  EncoderBlock  :87-130   (ln_1 -> self_attention -> +input ; ln_2 -> mlp -> +)
  Encoder       :133-175  (+pos_embedding, layers, final ln)
  VisionTransformer :178-351 (conv_proj patchify :308-333, class token :341-342, x[:,0] :347)
Parameter layout of the attention follows the forked ``MultiheadAttention``
(``utils.py:693-706``: packed ``in_proj_weight [3E,E]``, ``in_proj_bias [3E]``,
``out_proj.{weight,bias}``).

PARITY UNPINNED BY THE REFERENCE: ``vit.py`` cannot be imported here (torchvision
is absent) and its forward raises upstream in both train and eval mode
(``utils.py:210,219,227,877``).  The attention arithmetic is therefore
that of the third-party module it forks -- ``torch.nn.MultiheadAttention`` /
``F.multi_head_attention_forward`` (torch, reference pin ``torch>=1.10``
``setup.py:20``; installed 2.10.0) -- and the MLP is ``torchvision.ops.misc.MLP``
(pin ``torchvision==0.13.1``, ``setup.py:24``) = Linear,GELU,Dropout,Linear,Dropout.
``tests/test_oracle_vit.py`` pins this file against the installed
``torch.nn.MultiheadAttention`` + ``nn.LayerNorm(eps=1e-6)`` + ``nn.Conv2d``.
For ``robust=True`` the normalisation is ``SinkhornAttention`` (``utils.py:1031-1037``),
as wired in every runnable sibling (``simple_vit.py:56-57``, ``swin.py:240-244``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .simple_vit_oracle import _Q, _bf16, gelu_erf, layer_norm, sinkhorn_normalise

Tensor = torch.Tensor


def patchify_cp1p2(img: Tensor, p: int) -> Tensor:
    """Conv2d(k=s=p) as a matmul: features ordered (c, p1, p2) to match weight.reshape(D, -1)  (vit.py:237-242,323)."""
    b, c, H, W = img.shape
    h, w = H // p, W // p
    t = img.reshape(b, c, h, p, w, p).permute(0, 2, 4, 1, 3, 5)      # b h w c p1 p2
    return t.reshape(b, h * w, c * p * p)


def mha_self_attention(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
                       heads: int, robust: bool, Q: _Q, adrop=None) -> Tensor:
    """Self-attention of nn.MultiheadAttention(batch_first=True, need_weights=False).

    q,k,v = split(x W_in^T + b_in); per head softmax(q k^T / sqrt(dh)) v; concat heads; out_proj.
    `adrop` = (p, keep [B,H,S,S]): training-mode dropout on the attention weights (F.multi_head_attention_forward applies
    `dropout(attn)` between the softmax and attn @ v) with a given keep mask; fp32 path only.
    """
    B, S, E = x.shape
    dh = E // heads
    qkv = Q(x @ Q(in_w, "w").t() + in_b, "qkv")
    q, k, v = qkv.chunk(3, dim=-1)
    q, k, v = (t.reshape(B, S, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
    dots = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if not Q.on("p"):
        attn = torch.softmax(dots, dim=-1)
        if robust:
            attn = sinkhorn_normalise(attn)
        if adrop is not None:
            attn = attn * (adrop[1].to(attn.dtype) * (1.0 / (1.0 - adrop[0])))
        o = torch.matmul(attn, v)
    else:
        assert adrop is None, "attention dropout: fp32 oracle only"
        m = dots.max(dim=-1, keepdim=True).values
        p = torch.exp(dots - m)
        l = p.sum(dim=-1, keepdim=True)
        if robust:
            o = torch.matmul(_bf16(sinkhorn_normalise(p / l)), v)
        else:
            o = torch.matmul(_bf16(p), v) / l
    o = Q(o.permute(0, 2, 1, 3).reshape(B, S, E), "o")
    return o @ Q(out_w, "w").t() + out_b


def _drop(t: Tensor, drop, site: int) -> Tensor:
    """nn.Dropout in training mode with a GIVEN keep mask: t * keep / (1 - p) (vit.py:100-101,125,175).  `drop` = (p, keep) with
    keep(site, shape) -> mask tensor (nonzero = kept); None = no dropout.  The random stream of torch's own nn.Dropout is not part
    of the contract: parity is defined for injected masks."""
    if drop is None:
        return t
    p, keep = drop
    return t * (keep(site, tuple(t.shape)).to(t.dtype) * (1.0 / (1.0 - p)))


def encoder_block(x: Tensor, sd: Dict[str, Tensor], pfx: str, heads: int, robust: bool, Q: _Q,
                  eps: float = 1e-6, capture: Optional[dict] = None, tag: str = "", drop=None, layer: int = 0, attn_drop=None) -> Tensor:
    """EncoderBlock.forward, vit.py:118-130.  `capture[tag + ".attn_out"]` receives the residual stream
    after the attention half (vit.py:126), for the per-half localisation in tests/test_model_gpu.py.  Dropout sites of layer i:
    3 i the attention branch (vit.py:125), 3 i + 1 behind the GELU (vit.py:100), 3 i + 2 behind the second Linear (vit.py:101);
    the masks are over the flattened [batch * tokens, features] matrices."""
    a = Q(layer_norm(x, sd[pfx + "ln_1.weight"], sd[pfx + "ln_1.bias"], eps), "xn")
    Bn, S, D = x.shape
    adrop = None if attn_drop is None else (attn_drop[0], attn_drop[1](-(2 + layer), (Bn, heads, S, S)))      # site -(2 + layer), vit.py:108
    a = mha_self_attention(a, sd[pfx + "self_attention.in_proj_weight"], sd[pfx + "self_attention.in_proj_bias"],
                           sd[pfx + "self_attention.out_proj.weight"], sd[pfx + "self_attention.out_proj.bias"],
                           heads, robust, Q, adrop)
    a = _drop(a.reshape(Bn * S, D), drop, 3 * layer).reshape(Bn, S, D)
    x = a + x
    if capture is not None:
        capture[tag + ".attn_out"] = x.detach().clone()
    y = Q(layer_norm(x, sd[pfx + "ln_2.weight"], sd[pfx + "ln_2.bias"], eps), "xn")
    u = y @ Q(sd[pfx + "mlp.0.weight"], "w").t() + sd[pfx + "mlp.0.bias"]
    h = Q(gelu_erf(u), "h")
    h = _drop(h.reshape(Bn * S, -1), drop, 3 * layer + 1).reshape(Bn, S, -1)
    y = h @ Q(sd[pfx + "mlp.3.weight"], "w").t() + sd[pfx + "mlp.3.bias"]
    y = _drop(y.reshape(Bn * S, D), drop, 3 * layer + 2).reshape(Bn, S, D)
    return x + y


def vit_forward(sd: Dict[str, Tensor], img: Tensor, *, patch_size: int, num_heads: int,
                robust: bool = False, emulate_bf16=False, eps: float = 1e-6,
                capture: Optional[dict] = None, drop=None, attn_drop=None) -> Tensor:
    """VisionTransformer.forward, vit.py:335-351.  `drop` = (p, keep(site, shape)): training-mode dropout with given masks
    (site -1: the encoder input [B, S, D], vit.py:175; the blocks' sites: encoder_block); `attn_drop` = (p, keep): dropout on the attention
    weights of every block (attention_dropout, vit.py:108), keep(-(2 + layer), (B, H, S, S))."""
    Q = _Q(emulate_bf16)
    w = sd["conv_proj.weight"]
    D = w.shape[0]
    x = patchify_cp1p2(Q(img, "img"), patch_size) @ Q(w.reshape(D, -1), "w").t() + sd["conv_proj.bias"]
    B = x.shape[0]
    x = torch.cat([sd["class_token"].expand(B, -1, -1), x], dim=1)
    x = x + sd["encoder.pos_embedding"]
    x = _drop(x, drop, -1)
    if capture is not None:
        capture["embed"] = x.detach().clone()
    i = 0
    while f"encoder.layers.encoder_layer_{i}.ln_1.weight" in sd:
        x = encoder_block(x, sd, f"encoder.layers.encoder_layer_{i}.", num_heads, robust, Q, eps, capture, f"layer{i}", drop, i, attn_drop)
        if capture is not None:
            capture[f"layer{i}.out"] = x.detach().clone()
        i += 1
    x = layer_norm(x, sd["encoder.ln.weight"], sd["encoder.ln.bias"], eps)
    x = x[:, 0]
    if "heads.pre_logits.weight" in sd:
        x = torch.tanh(x @ sd["heads.pre_logits.weight"].t() + sd["heads.pre_logits.bias"])
    return x @ sd["heads.head.weight"].t() + sd["heads.head.bias"]


def vit_init_state_dict(*, image_size: int, patch_size: int, num_layers: int, num_heads: int,
                        hidden_dim: int, mlp_dim: int, num_classes: int = 1000, seed: int = 0,
                        randomise_head: bool = True) -> Dict[str, Tensor]:
    """Fresh weights following the reference initialisers (vit.py:49-53,151-153,247,273-306; utils.py:718-732).

    ``randomise_head`` replaces the zero-init ``heads.head`` (vit.py:304-306) by N(0, 0.02) so logits are
    non-trivial in parity tests.
    """
    g = torch.Generator().manual_seed(seed)
    D, M = hidden_dim, mlp_dim
    S = (image_size // patch_size) ** 2 + 1
    sd: Dict[str, Tensor] = {}

    def xavier(out_f, in_f):
        a = math.sqrt(6.0 / (in_f + out_f))
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * a

    fan_in = 3 * patch_size * patch_size
    wconv = torch.empty(D, 3, patch_size, patch_size)
    torch.nn.init.trunc_normal_(wconv, std=math.sqrt(1 / fan_in), generator=g)
    sd["conv_proj.weight"] = wconv
    sd["conv_proj.bias"] = torch.zeros(D)
    sd["class_token"] = torch.zeros(1, 1, D)
    sd["encoder.pos_embedding"] = torch.randn(1, S, D, generator=g) * 0.02
    for i in range(num_layers):
        p = f"encoder.layers.encoder_layer_{i}."
        sd[p + "ln_1.weight"] = torch.ones(D)
        sd[p + "ln_1.bias"] = torch.zeros(D)
        sd[p + "self_attention.in_proj_weight"] = xavier(3 * D, D)
        sd[p + "self_attention.in_proj_bias"] = torch.zeros(3 * D)
        bound = 1.0 / math.sqrt(D)      # nn.Linear default (kaiming_uniform a=sqrt(5))
        sd[p + "self_attention.out_proj.weight"] = (torch.rand(D, D, generator=g) * 2 - 1) * bound
        sd[p + "self_attention.out_proj.bias"] = torch.zeros(D)
        sd[p + "ln_2.weight"] = torch.ones(D)
        sd[p + "ln_2.bias"] = torch.zeros(D)
        sd[p + "mlp.0.weight"] = xavier(M, D)
        sd[p + "mlp.0.bias"] = torch.randn(M, generator=g) * 1e-6
        sd[p + "mlp.3.weight"] = xavier(D, M)
        sd[p + "mlp.3.bias"] = torch.randn(D, generator=g) * 1e-6
    sd["encoder.ln.weight"] = torch.ones(D)
    sd["encoder.ln.bias"] = torch.zeros(D)
    if randomise_head:
        sd["heads.head.weight"] = torch.randn(num_classes, D, generator=g) * 0.02
        sd["heads.head.bias"] = torch.randn(num_classes, generator=g) * 0.02
    else:
        sd["heads.head.weight"] = torch.zeros(num_classes, D)
        sd["heads.head.bias"] = torch.zeros(num_classes)
    return sd
