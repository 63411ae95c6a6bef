"""CPU: pins the VisionTransformer oracle (oracle/vit_oracle.py) against the third-party arithmetic the reference
forks -- torch.nn.MultiheadAttention / nn.LayerNorm(eps=1e-6) / nn.Conv2d patchify -- because the reference's own
vit.py cannot be imported (torchvision absent) and its forward raises upstream.
"parity unpinned by the reference": there is no reference fixture for this path.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from oracle import vit_oracle as V


class TorchVT(nn.Module):
    """The module tree of vit.py:178-351 built from stock torch layers (what the reference intends to compute)."""

    def __init__(self, image_size, patch_size, num_layers, num_heads, hidden_dim, mlp_dim, num_classes):
        super().__init__()
        self.p, self.D = patch_size, hidden_dim
        self.conv_proj = nn.Conv2d(3, hidden_dim, patch_size, patch_size)
        S = (image_size // patch_size) ** 2 + 1
        self.class_token = nn.Parameter(torch.zeros(1, 1, hidden_dim))
        self.pos = nn.Parameter(torch.zeros(1, S, hidden_dim))
        self.blocks = nn.ModuleList()
        for _ in range(num_layers):
            self.blocks.append(nn.ModuleDict(dict(
                ln_1=nn.LayerNorm(hidden_dim, eps=1e-6),
                attn=nn.MultiheadAttention(hidden_dim, num_heads, batch_first=True),
                ln_2=nn.LayerNorm(hidden_dim, eps=1e-6),
                fc1=nn.Linear(hidden_dim, mlp_dim), fc2=nn.Linear(mlp_dim, hidden_dim))))
        self.ln = nn.LayerNorm(hidden_dim, eps=1e-6)
        self.head = nn.Linear(hidden_dim, num_classes)

    def forward(self, x, drop=None):
        """`drop` = (p, keep(site, shape)): nn.Dropout in training mode (vit.py:100-101,125,175) with GIVEN keep masks, as a
        multiplication with keep / (1 - p) where the reference has its Dropout modules."""
        def dr(t, site):
            if drop is None:
                return t
            p, keep = drop
            return (t.reshape(-1, t.shape[-1]) * (keep(site, (t.numel() // t.shape[-1], t.shape[-1])).to(t.dtype) / (1 - p))).reshape(t.shape) \
                if site >= 0 else t * (keep(site, tuple(t.shape)).to(t.dtype) / (1 - p))
        n = x.shape[0]
        x = self.conv_proj(x).reshape(n, self.D, -1).permute(0, 2, 1)
        x = dr(torch.cat([self.class_token.expand(n, -1, -1), x], dim=1) + self.pos, -1)
        for i, b in enumerate(self.blocks):
            y = b["ln_1"](x)
            y, _ = b["attn"](y, y, y, need_weights=False)
            x = x + dr(y, 3 * i)
            y = dr(b["fc2"](dr(torch.nn.functional.gelu(b["fc1"](b["ln_2"](x))), 3 * i + 1)), 3 * i + 2)
            x = x + y
        return self.head(self.ln(x)[:, 0])


def load_into_torch(m: TorchVT, sd):
    with torch.no_grad():
        m.conv_proj.weight.copy_(sd["conv_proj.weight"]); m.conv_proj.bias.copy_(sd["conv_proj.bias"])
        m.class_token.copy_(sd["class_token"]); m.pos.copy_(sd["encoder.pos_embedding"])
        for i, b in enumerate(m.blocks):
            p = f"encoder.layers.encoder_layer_{i}."
            b["ln_1"].weight.copy_(sd[p + "ln_1.weight"]); b["ln_1"].bias.copy_(sd[p + "ln_1.bias"])
            b["attn"].in_proj_weight.copy_(sd[p + "self_attention.in_proj_weight"])
            b["attn"].in_proj_bias.copy_(sd[p + "self_attention.in_proj_bias"])
            b["attn"].out_proj.weight.copy_(sd[p + "self_attention.out_proj.weight"])
            b["attn"].out_proj.bias.copy_(sd[p + "self_attention.out_proj.bias"])
            b["ln_2"].weight.copy_(sd[p + "ln_2.weight"]); b["ln_2"].bias.copy_(sd[p + "ln_2.bias"])
            b["fc1"].weight.copy_(sd[p + "mlp.0.weight"]); b["fc1"].bias.copy_(sd[p + "mlp.0.bias"])
            b["fc2"].weight.copy_(sd[p + "mlp.3.weight"]); b["fc2"].bias.copy_(sd[p + "mlp.3.bias"])
        m.ln.weight.copy_(sd["encoder.ln.weight"]); m.ln.bias.copy_(sd["encoder.ln.bias"])
        m.head.weight.copy_(sd["heads.head.weight"]); m.head.bias.copy_(sd["heads.head.bias"])


def test_vit_oracle_matches_torch_multihead_attention_stack():
    cfg = dict(image_size=64, patch_size=16, num_layers=2, num_heads=3, hidden_dim=96, mlp_dim=192, num_classes=11)
    sd = V.vit_init_state_dict(seed=1, **cfg)
    g = torch.Generator().manual_seed(2)
    for k in sd:                                    # make the zero-initialised tensors non-trivial
        if k.endswith("bias") or k == "class_token":
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    m = TorchVT(**cfg).eval()
    load_into_torch(m, sd)
    x = torch.randn(3, 3, 64, 64, generator=g)
    ref = m(x)
    out = V.vit_forward(sd, x, patch_size=16, num_heads=3)
    assert (out - ref).abs().max() / ref.abs().max() < 5e-6


def make_keep(seed: int, p: float):
    """A reproducible family of keep masks: site -> Bernoulli(1 - p) mask of the asked shape (same seed, same masks)."""
    def keep(site, shape):
        g = torch.Generator().manual_seed(seed * 1000 + site + 7)
        return (torch.rand(shape, generator=g) >= p).to(torch.uint8)
    return keep


def test_vit_oracle_dropout_sites_match_the_torch_module_tree():
    """Training-mode dropout with injected masks: the oracle's sites (encoder input, attention branch, behind the GELU, behind the
    second Linear) are where the stock-layer tree has its Dropout modules; logits and gradients agree."""
    cfg = dict(image_size=64, patch_size=16, num_layers=2, num_heads=3, hidden_dim=96, mlp_dim=192, num_classes=11)
    sd = V.vit_init_state_dict(seed=3, **cfg)
    g = torch.Generator().manual_seed(4)
    for k in sd:
        if k.endswith("bias") or k == "class_token":
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    m = TorchVT(**cfg)
    load_into_torch(m, sd)
    x = torch.randn(3, 3, 64, 64, generator=g)
    drop = (0.25, make_keep(5, 0.25))
    ref = m(x, drop)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = V.vit_forward(sdg, x, patch_size=16, num_heads=3, drop=drop)
    assert (out - ref).abs().max() / ref.abs().max() < 5e-6
    assert (out - V.vit_forward(sd, x, patch_size=16, num_heads=3)).abs().max() > 1e-3      # the masks do something
    out.square().sum().backward()
    ref.square().sum().backward()
    w = sdg["encoder.layers.encoder_layer_0.mlp.0.weight"].grad
    assert (w - m.blocks[0]["fc1"].weight.grad).abs().max() < 1e-5 * w.abs().max() + 1e-7


def test_vit_oracle_attention_dropout_is_dropout_of_the_attention_weights():
    """attention_dropout (vit.py:108 -> nn.MultiheadAttention(dropout=p)): torch's module draws its own mask, so the oracle's site is
    pinned against the definition written out -- out_proj(concat_h((softmax(q k^T / sqrt(dh)) * keep / (1 - p)) v))."""
    torch.manual_seed(0)
    B, S, E, H = 2, 5, 24, 3
    x = torch.randn(B, S, E)
    in_w, in_b = torch.randn(3 * E, E) * 0.2, torch.randn(3 * E) * 0.1
    out_w, out_b = torch.randn(E, E) * 0.2, torch.randn(E) * 0.1
    p = 0.3
    keep = (torch.rand(B, H, S, S) >= p).to(torch.uint8)
    got = V.mha_self_attention(x, in_w, in_b, out_w, out_b, H, False, V._Q(False), (p, keep))
    q, k, v = (x @ in_w.t() + in_b).chunk(3, dim=-1)
    sp = lambda t: t.reshape(B, S, H, E // H).transpose(1, 2)
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / (E // H) ** 0.5, dim=-1) * keep / (1 - p)
    ref = (a @ sp(v)).transpose(1, 2).reshape(B, S, E) @ out_w.t() + out_b
    assert (got - ref).abs().max() < 1e-5
    m = nn.MultiheadAttention(E, H, batch_first=True).eval()          # and with keep = 1 it is torch's module
    with torch.no_grad():
        m.in_proj_weight.copy_(in_w); m.in_proj_bias.copy_(in_b); m.out_proj.weight.copy_(out_w); m.out_proj.bias.copy_(out_b)
    ones = torch.ones(B, H, S, S, dtype=torch.uint8)
    full = V.mha_self_attention(x, in_w, in_b, out_w, out_b, H, False, V._Q(False), (0.0, ones))
    assert (full - m(x, x, x, need_weights=False)[0]).abs().max() < 1e-5


def test_patchify_cp1p2_is_conv2d():
    g = torch.Generator().manual_seed(0)
    conv = nn.Conv2d(3, 8, 16, 16)
    x = torch.randn(2, 3, 32, 48, generator=g)
    ref = conv(x).reshape(2, 8, -1).permute(0, 2, 1)
    out = V.patchify_cp1p2(x, 16) @ conv.weight.reshape(8, -1).t() + conv.bias
    assert (out - ref).abs().max() < 1e-5


def test_init_follows_reference_initialisers():
    sd = V.vit_init_state_dict(image_size=32, patch_size=16, num_layers=1, num_heads=2, hidden_dim=64, mlp_dim=128,
                               num_classes=5, randomise_head=False)
    assert sd["heads.head.weight"].abs().max() == 0 and sd["class_token"].abs().max() == 0      # vit.py:247,304-306
    assert sd["encoder.layers.encoder_layer_0.self_attention.in_proj_bias"].abs().max() == 0     # utils.py:727
    assert abs(sd["encoder.pos_embedding"].std().item() - 0.02) < 0.005                          # vit.py:151-153
    assert sd["encoder.layers.encoder_layer_0.mlp.0.bias"].abs().max() < 1e-4                    # vit.py:53
