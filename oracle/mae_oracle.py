"""Oracle: CPU restatement of the reference MAE wrapper and of the lucidrains-style ViT it can wrap (test infrastructure only).

Follows ``/root/reference/vit_pytorch_robust/mae.py:51-118`` (forward) and, for the encoder / decoder transformer,
``learnable_memory_vit.py:30-104`` (FeedForward :30-42 with keys ``net.0,1,4``; Attention :44-85 with ``to_q``, ``to_kv``
bias-free and biased ``to_out.0``; Transformer :87-104, no final norm).

Pinning note: ``mae.py:6`` imports ``Transformer`` from ``vit.py``, which defines none, so the reference MAE is importable
only with a shim: ``tests/golden/gen_golden_mae.py`` seeds ``sys.modules['vit_pytorch_robust.vit']``
with a module whose ``Transformer`` is ``learnable_memory_vit.Transformer`` and wraps ``learnable_memory_vit.ViT``.
The fixture ``tests/golden/mae_small.npz`` ("reference MAE + shimmed missing import") pins this file.
"""
from __future__ import annotations

from typing import Dict

import torch

from .simple_vit_oracle import _Q, _bf16, gelu_erf, layer_norm, patchify_p1p2c

Tensor = torch.Tensor


def _drop(t: Tensor, drop, site: int, shape=None) -> Tensor:
    """nn.Dropout in training mode with a GIVEN keep mask (learnable_memory_vit.py:37,39,54,61,83): t * keep / (1 - p); `drop` = (p, keep(site, shape))."""
    if drop is None:
        return t
    p, keep = drop
    shp = tuple(t.shape) if shape is None else shape
    return t * (keep(site, shp).reshape(t.shape).to(t.dtype) * (1.0 / (1.0 - p)))


def lucid_attention(x: Tensor, sd: Dict[str, Tensor], pfx: str, heads: int, dim_head: int, Q: _Q, drop=None, layer: int = 0) -> Tensor:
    """Dropout sites of layer i as in vit_oracle.encoder_block: -(2 + i) the attention weights [B,H,N,N] (:83), 3 i the to_out dropout (:61)."""
    B, N, _ = x.shape
    xn = Q(layer_norm(x, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], 1e-5))
    q = Q(xn @ Q(sd[pfx + "to_q.weight"]).t())
    kv = Q(xn @ Q(sd[pfx + "to_kv.weight"]).t())
    k, v = kv.chunk(2, dim=-1)
    q, k, v = (t.reshape(B, N, heads, dim_head).permute(0, 2, 1, 3) for t in (q, k, v))
    dots = torch.matmul(q, k.transpose(-1, -2)) * dim_head ** -0.5
    if Q.emulate:
        m = dots.max(dim=-1, keepdim=True).values
        p = torch.exp(dots - m)
        o = torch.matmul(_bf16(p), v) / p.sum(dim=-1, keepdim=True)
    else:
        assert drop is None or not Q.emulate
        o = torch.matmul(_drop(torch.softmax(dots, dim=-1), drop, -(2 + layer)), v)
    o = Q(o.permute(0, 2, 1, 3).reshape(B, N, heads * dim_head))
    y = o @ Q(sd[pfx + "to_out.0.weight"]).t() + sd[pfx + "to_out.0.bias"]
    return _drop(y, drop, 3 * layer, (B * N, y.shape[-1]))


def lucid_feed_forward(x: Tensor, sd: Dict[str, Tensor], pfx: str, Q: _Q, drop=None, layer: int = 0) -> Tensor:
    """Dropout sites 3 i + 1 behind the GELU (:37) and 3 i + 2 behind the second Linear (:39)."""
    B, N, _ = x.shape
    xn = Q(layer_norm(x, sd[pfx + "net.0.weight"], sd[pfx + "net.0.bias"], 1e-5))
    h = Q(gelu_erf(xn @ Q(sd[pfx + "net.1.weight"]).t() + sd[pfx + "net.1.bias"]))
    h = _drop(h, drop, 3 * layer + 1, (B * N, h.shape[-1]))
    y = h @ Q(sd[pfx + "net.4.weight"]).t() + sd[pfx + "net.4.bias"]
    return _drop(y, drop, 3 * layer + 2, (B * N, y.shape[-1]))


def lucid_transformer(x: Tensor, sd: Dict[str, Tensor], pfx: str, heads: int, dim_head: int, Q: _Q, drop=None) -> Tensor:
    """`drop` = (p, keep(site, shape)): the one `dropout` of learnable_memory_vit.Transformer (:90-96) at its four sites per layer."""
    i = 0
    while f"{pfx}layers.{i}.0.norm.weight" in sd:
        x = lucid_attention(x, sd, f"{pfx}layers.{i}.0.", heads, dim_head, Q, drop, i) + x
        x = lucid_feed_forward(x, sd, f"{pfx}layers.{i}.1.", Q, drop, i) + x
        i += 1
    return x


def mae_forward(sd: Dict[str, Tensor], img: Tensor, rand_indices: Tensor, *, patch_size: int, enc_heads: int,
                dec_heads: int, masking_ratio: float = 0.75, dim_head: int = 64, emulate_bf16: bool = False) -> Tensor:
    """MAE.forward (mae.py:51-118) with the random permutation given explicitly (mae.py:67)."""
    Q = _Q(emulate_bf16)
    patches = patchify_p1p2c(img, patch_size, patch_size)                       # to_patch  (b (h w) (p1 p2 c))
    B, n, _ = patches.shape
    tokens = Q(patches) @ Q(sd["encoder.to_patch_embedding.1.weight"]).t() + sd["encoder.to_patch_embedding.1.bias"]
    tokens = tokens + sd["encoder.pos_embedding"][:, 1:n + 1]
    num_masked = int(masking_ratio * n)
    masked, unmasked = rand_indices[:, :num_masked], rand_indices[:, num_masked:]
    br = torch.arange(B)[:, None]
    tokens = tokens[br, unmasked]
    masked_patches = patches[br, masked]
    enc = lucid_transformer(tokens, sd, "encoder.transformer.", enc_heads, dim_head, Q)
    if "enc_to_dec.weight" in sd:
        dec = Q(enc) @ Q(sd["enc_to_dec.weight"]).t() + sd["enc_to_dec.bias"]
    else:
        dec = enc
    demb = sd["decoder_pos_emb.weight"]
    dec = dec + demb[unmasked]
    mask_tokens = sd["mask_token"][None, None, :].expand(B, num_masked, -1) + demb[masked]
    full = torch.zeros(B, n, dec.shape[-1], dtype=dec.dtype)
    full[br, unmasked] = dec
    full[br, masked] = mask_tokens
    out = lucid_transformer(full, sd, "decoder.", dec_heads, dim_head, Q)
    pred = Q(out[br, masked]) @ Q(sd["to_pixels.weight"]).t() + sd["to_pixels.bias"]
    return torch.nn.functional.mse_loss(pred, masked_patches)
