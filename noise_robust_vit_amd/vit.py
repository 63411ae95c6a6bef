"""Drop-in torchvision-style `VisionTransformer` on the MI355X HIP hot path.

Mirrors the reference's module tree, constructor arguments, initialisers and state_dict keys
(`/root/reference/vit_pytorch_robust/vit.py`: MLPBlock :35-84, EncoderBlock :87-130, Encoder :133-175,
VisionTransformer :178-351, builders :354-519; packed attention parameters `utils.py:693-706`, init
`utils.py:718-732`).  Unlike the reference it does not need torchvision and its forward actually runs
(upstream's raises): the attention arithmetic is that of `torch.nn.MultiheadAttention`
self-attention, fused in `libnrv_hip.so`.

As in `simple_vit.py`, the nn.Linear / nn.LayerNorm / nn.Conv2d children only hold parameters.
Dropout arguments are accepted; the fused path implements p = 0 (the reference defaults, vit.py:189-190).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from functools import partial
from typing import Any, Callable, Optional

import torch
import torch.nn as nn

from ._lib import PATCH_CP1P2, NrvError
from .encoder import AttnHalfFn, BlockMeta, EncoderStackFn, MlpHalfFn, PatchEmbedFn
from .simple_vit import SinkhornAttention

__all__ = ["VisionTransformer", "Encoder", "EncoderBlock", "MLPBlock", "MultiheadAttention", "interpolate_embeddings",
           "vit_s_16", "vit_b_16", "vit_b_32", "vit_l_16", "vit_l_32", "vit_h_14"]



def _input_dropout(x: torch.Tensor, meta: BlockMeta, p: float) -> torch.Tensor:
    """Encoder-level dropout on the embedded tokens (vit.py:154,175): once per step, plain tensor ops.  Site -1 of the mask source."""
    if meta.mask_source is not None:
        keep = meta.mask_source(-1, tuple(x.shape)).to(x.device)
    else:
        keep = torch.rand(x.shape, device=x.device) >= p
    return x * (keep.to(x.dtype) * (1.0 / (1.0 - p)))


class MultiheadAttention(nn.Module):
    """Parameter layout and initialisation of the reference's forked nn.MultiheadAttention (utils.py:650-732):
    packed `in_proj_weight [3E, E]`, `in_proj_bias [3E]`, `out_proj` Linear(E, E) with bias.

    `forward(x, x, x, need_weights=False)` runs fused self-attention and returns `(out, None)` like the
    reference's call site expects (vit.py:124); `need_weights=True` also returns the attention weights (recomputed, no
    gradient).  Only the self-attention form used by `EncoderBlock` exists; masks and attention dropout run on the composed path.
    """

    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None, batch_first=False, device=None, dtype=None, robust=False) -> None:
        super().__init__()
        if not bias or add_bias_kv or add_zero_attn or (kdim not in (None, embed_dim)) or (vdim not in (None, embed_dim)):
            raise NotImplementedError("only the packed self-attention configuration used by EncoderBlock is supported")
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.batch_first, self.robust = batch_first, robust
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.attend = SinkhornAttention(-1) if robust else None
        nn.init.xavier_uniform_(self.in_proj_weight)          # utils.py:718-732
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def attn_params(self):
        return [self.in_proj_weight, self.in_proj_bias, self.out_proj.weight, self.out_proj.bias]

    def _score_bias(self, x, attn_mask, key_padding_mask):
        """torch's mask semantics (utils.py:741-751 -> F.multi_head_attention_forward) as ONE additive fp32 bias broadcastable to
        [B, H, N, N]: `attn_mask` [N, N] or [B * H, N, N], bool (True = not allowed) or float (added to the scores);
        `key_padding_mask` [B, N], bool (True = ignore that key) or float.  None when no mask is given (the fused kernels run)."""
        if attn_mask is None and key_padding_mask is None:
            return None
        B, N = x.shape[0], x.shape[1]
        H = self.num_heads
        bias = torch.zeros(B, H, N, N, dtype=torch.float32, device=x.device)
        for m, view in ((attn_mask, None), (key_padding_mask, "keys")):
            if m is None:
                continue
            m = m.to(x.device)
            add = torch.zeros(m.shape, dtype=torch.float32, device=x.device).masked_fill_(m, float("-inf")) if m.dtype == torch.bool else m.to(torch.float32)
            if view == "keys":
                if tuple(add.shape) != (B, N):
                    raise ValueError(f"key_padding_mask must be [{B}, {N}], got {tuple(add.shape)}")
                bias += add[:, None, None, :]
            elif add.dim() == 2:
                if tuple(add.shape) != (N, N):
                    raise ValueError(f"attn_mask must be [{N}, {N}] or [{B * H}, {N}, {N}], got {tuple(add.shape)}")
                bias += add
            else:
                if tuple(add.shape) != (B * H, N, N):
                    raise ValueError(f"attn_mask must be [{N}, {N}] or [{B * H}, {N}, {N}], got {tuple(add.shape)}")
                bias += add.reshape(B, H, N, N)
        return bias

    def forward(self, query, key=None, value=None, key_padding_mask=None, need_weights=False, attn_mask=None, **kw):
        """Self-attention `forward(x, x, x, need_weights=False) -> (out, None)` (utils.py:741-751,594-597; the call site is
        vit.py:124): packed QKV projection + bias -> fused softmax (or Sinkhorn, robust=True) attention -> out_proj + bias,
        all in libnrv_hip.so.  `batch_first=False` takes / returns [S, B, E] like torch's module."""
        if (key is not None and key is not query) or (value is not None and value is not query):
            raise NotImplementedError("self-attention only (query is key is value), as EncoderBlock calls it")
        x = query if self.batch_first else query.transpose(0, 1)
        meta = BlockMeta(heads=self.num_heads, dim_head=self.head_dim, eps=0.0, robust=bool(self.robust),
                         attn_dropout=self.dropout if self.training else 0.0, mask_source=getattr(self, "mask_source", None),
                         attn_bias=self._score_bias(x, attn_mask, key_padding_mask))
        if not need_weights:
            out = AttnHalfFn.apply(x, meta, None, None, *self.attn_params())
            return (out if self.batch_first else out.transpose(0, 1)), None
        # need_weights=True (utils.py:741-751, 594-597): the fused kernels never materialise the weights, so they are recomputed from
        # q, k and the saved statistics (nrv_attn_probs; the Sinkhorn-scaled matrix for robust=True) -- [B, N, N] averaged over the
        # heads as torch's module returns them by default, [B, H, N, N] with average_attn_weights=False; introspection: no gradient
        if meta.attn_bias is not None or meta.attn_dropout > 0.0:
            raise NotImplementedError("need_weights=True together with masks or attention dropout")
        from .encoder import record_attention
        maps: list = []
        with record_attention(maps):
            out = AttnHalfFn.apply(x, meta, None, None, *self.attn_params())
        w = maps[-1]
        if kw.get("average_attn_weights", True):
            w = w.mean(dim=1)
        return (out if self.batch_first else out.transpose(0, 1)), w


class MLPBlock(nn.Sequential):
    """Linear, GELU, Dropout, Linear, Dropout (torchvision.ops.misc.MLP as specialised at vit.py:35-53)."""

    _version = 2

    def __init__(self, in_dim: int, mlp_dim: int, dropout: float):
        super().__init__(nn.Linear(in_dim, mlp_dim), nn.GELU(), nn.Dropout(dropout),
                         nn.Linear(mlp_dim, in_dim), nn.Dropout(dropout))
        self.dropout_p = dropout
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.normal_(m.bias, std=1e-6)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # legacy checkpoints name the two Linears linear_1 / linear_2 (vit.py:55-84)
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for i in range(2):
                for kind in ("weight", "bias"):
                    old, new = f"{prefix}linear_{i + 1}.{kind}", f"{prefix}{3 * i}.{kind}"
                    if old in state_dict:
                        state_dict[new] = state_dict.pop(old)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def mlp_params(self):
        return [self[0].weight, self[0].bias, self[3].weight, self[3].bias]


class EncoderBlock(nn.Module):
    def __init__(self, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float, attention_dropout: float,
                 norm_layer: Callable[..., nn.Module] = partial(nn.LayerNorm, eps=1e-6), robust=False):
        super().__init__()
        self.num_heads = num_heads
        self.ln_1 = norm_layer(hidden_dim)
        self.self_attention = MultiheadAttention(hidden_dim, num_heads, dropout=attention_dropout,
                                                 batch_first=True, robust=robust)
        self.dropout = nn.Dropout(dropout)
        self.ln_2 = norm_layer(hidden_dim)
        self.mlp = MLPBlock(hidden_dim, mlp_dim, dropout)
        self._meta = BlockMeta(heads=num_heads, dim_head=hidden_dim // num_heads, eps=self.ln_1.eps, robust=bool(robust))

    def layer_params(self):
        return ([self.ln_1.weight, self.ln_1.bias] + self.self_attention.attn_params()
                + [self.ln_2.weight, self.ln_2.bias] + self.mlp.mlp_params())

    def forward(self, input: torch.Tensor):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        self._meta.dropout = self.dropout.p if self.training else 0.0          # vit.py:100-101,125
        self._meta.attn_dropout = self.self_attention.dropout if self.training else 0.0      # vit.py:108
        return EncoderStackFn.apply(input, self._meta, *self.layer_params())


class Encoder(nn.Module):
    def __init__(self, seq_length: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int,
                 dropout: float, attention_dropout: float,
                 norm_layer: Callable[..., nn.Module] = partial(nn.LayerNorm, eps=1e-6), robust=False):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.dropout = nn.Dropout(dropout)
        layers: "OrderedDict[str, nn.Module]" = OrderedDict()
        for i in range(num_layers):
            layers[f"encoder_layer_{i}"] = EncoderBlock(num_heads, hidden_dim, mlp_dim, dropout, attention_dropout,
                                                        norm_layer, robust=robust)
        self.layers = nn.Sequential(layers)
        self.ln = norm_layer(hidden_dim)
        self._meta = BlockMeta(heads=num_heads, dim_head=hidden_dim // num_heads, eps=self.ln.eps, robust=bool(robust))

    def attach_grad_sink(self, sink) -> None:
        self._meta.sink = sink

    def run_stack(self, x: torch.Tensor) -> torch.Tensor:
        """All encoder layers in one autograd node (positional embedding already added, no final LN)."""
        flat = []
        for blk in self.layers:
            flat += blk.layer_params()
        p = self.dropout.p if self.training else 0.0
        self._meta.attn_dropout = self.layers[0].self_attention.dropout if self.training and len(self.layers) else 0.0      # vit.py:108, one value for all blocks (vit.py:161)
        self._meta.dropout = p                                      # the blocks' MLP / branch dropout (vit.py:100-101,125): same p (vit.py:161)
        if p > 0.0:
            x = _input_dropout(x, self._meta, p)                    # vit.py:175
        return EncoderStackFn.apply(x, self._meta, *flat)

    def forward(self, input: torch.Tensor):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        x = self.run_stack(input + self.pos_embedding)
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), self.ln.weight, self.ln.bias, self.ln.eps)


class VisionTransformer(nn.Module):
    """Vision Transformer as per https://arxiv.org/abs/2010.11929 (reference signature, vit.py:181-196)."""

    def __init__(self, image_size: int, patch_size: int, num_layers: int, num_heads: int, hidden_dim: int,
                 mlp_dim: int, dropout: float = 0.0, attention_dropout: float = 0.0, num_classes: int = 1000,
                 representation_size: Optional[int] = None,
                 norm_layer: Callable[..., nn.Module] = partial(nn.LayerNorm, eps=1e-6),
                 conv_stem_configs=None, robust: bool = False):
        super().__init__()
        torch._assert(image_size % patch_size == 0, "Input shape indivisible by patch size!")
        if conv_stem_configs is not None:
            raise NotImplementedError("conv-stem variants are outside the ViT-S/B/L encoder hot path")
        self.image_size, self.patch_size, self.hidden_dim, self.mlp_dim = image_size, patch_size, hidden_dim, mlp_dim
        self.attention_dropout, self.dropout, self.num_classes = attention_dropout, dropout, num_classes
        self.representation_size, self.norm_layer, self.robust = representation_size, norm_layer, robust

        self.conv_proj = nn.Conv2d(in_channels=3, out_channels=hidden_dim, kernel_size=patch_size, stride=patch_size)
        seq_length = (image_size // patch_size) ** 2
        self.class_token = nn.Parameter(torch.zeros(1, 1, hidden_dim))
        seq_length += 1
        self.encoder = Encoder(seq_length, num_layers, num_heads, hidden_dim, mlp_dim, dropout, attention_dropout,
                               norm_layer, robust=robust)
        self.seq_length = seq_length

        heads_layers: "OrderedDict[str, nn.Module]" = OrderedDict()
        if representation_size is None:
            heads_layers["head"] = nn.Linear(hidden_dim, num_classes)
        else:
            heads_layers["pre_logits"] = nn.Linear(hidden_dim, representation_size)
            heads_layers["act"] = nn.Tanh()
            heads_layers["head"] = nn.Linear(representation_size, num_classes)
        self.heads = nn.Sequential(heads_layers)

        # initialisers of vit.py:273-306
        fan_in = self.conv_proj.in_channels * self.conv_proj.kernel_size[0] * self.conv_proj.kernel_size[1]
        nn.init.trunc_normal_(self.conv_proj.weight, std=math.sqrt(1 / fan_in))
        nn.init.zeros_(self.conv_proj.bias)
        if hasattr(self.heads, "pre_logits"):
            nn.init.trunc_normal_(self.heads.pre_logits.weight, std=math.sqrt(1 / self.heads.pre_logits.in_features))
            nn.init.zeros_(self.heads.pre_logits.bias)
        nn.init.zeros_(self.heads.head.weight)
        nn.init.zeros_(self.heads.head.bias)
        self._sink = None

    def attach_grad_sink(self, sink) -> None:
        self._sink = sink
        self.encoder.attach_grad_sink(sink)

    def _process_input(self, x: torch.Tensor) -> torch.Tensor:
        """Patch tokens WITHOUT class token / positions (reference helper, vit.py:308-333), fp32 [n, n_h*n_w, D]."""
        n, c, h, w = x.shape
        torch._assert(h == self.image_size, f"Wrong image height! Expected {self.image_size} but got {h}!")
        torch._assert(w == self.image_size, f"Wrong image width! Expected {self.image_size} but got {w}!")
        zeros = torch.zeros((h // self.patch_size) * (w // self.patch_size), self.hidden_dim, device=x.device)
        return PatchEmbedFn.apply(x, self.conv_proj.weight, self.conv_proj.bias, zeros, None,
                                  self.patch_size, PATCH_CP1P2, None)

    def forward(self, x: torch.Tensor):
        n, c, h, w = x.shape
        torch._assert(h == self.image_size, f"Wrong image height! Expected {self.image_size} but got {h}!")
        torch._assert(w == self.image_size, f"Wrong image width! Expected {self.image_size} but got {w}!")
        # conv_proj + class token + pos_embedding in one GEMM epilogue (vit.py:323-342, Encoder :174)
        tokens = PatchEmbedFn.apply(x, self.conv_proj.weight, self.conv_proj.bias, self.encoder.pos_embedding,
                                    self.class_token, self.patch_size, PATCH_CP1P2, self._sink)
        tokens = self.encoder.run_stack(tokens)
        # encoder.ln is per token and only the class token is read (vit.py:175,347): normalise that row only
        cls = tokens[:, 0]
        ln = self.encoder.ln
        cls = torch.nn.functional.layer_norm(cls, (cls.shape[-1],), ln.weight, ln.bias, ln.eps)
        return self.heads(cls)


def interpolate_embeddings(image_size: int, patch_size: int, model_state, interpolation_mode: str = "bicubic",
                           reset_heads: bool = False):
    """Resize `encoder.pos_embedding` of a checkpoint to a new image size (same contract as vit.py:522-603): the class
    token's position is kept, the patch positions are viewed as a square grid and resampled (align_corners=True).
    Updates `model_state` in place and returns it (a copy without `heads.*` when `reset_heads`)."""
    pos = model_state["encoder.pos_embedding"]
    if pos.dim() != 3 or pos.shape[0] != 1:
        raise ValueError(f"Unexpected position embedding shape: {pos.shape}")
    hidden = pos.shape[2]
    old_tokens = pos.shape[1] - 1
    new_side = image_size // patch_size
    if new_side * new_side == old_tokens:
        return model_state
    old_side = math.isqrt(old_tokens)
    if old_side * old_side != old_tokens:
        raise ValueError(f"seq_length is not a perfect square! Instead got seq_length = {old_tokens}")
    grid = pos[:, 1:, :].transpose(1, 2).reshape(1, hidden, old_side, old_side)
    grid = nn.functional.interpolate(grid, size=new_side, mode=interpolation_mode, align_corners=True)
    grid = grid.reshape(1, hidden, new_side * new_side).transpose(1, 2)
    model_state["encoder.pos_embedding"] = torch.cat([pos[:, :1, :], grid], dim=1)
    if reset_heads:
        return OrderedDict((k, v) for k, v in model_state.items() if not k.startswith("heads"))
    return model_state


def _vision_transformer(patch_size: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int,
                        **kwargs: Any) -> VisionTransformer:
    image_size = kwargs.pop("image_size", 224)
    return VisionTransformer(image_size=image_size, patch_size=patch_size, num_layers=num_layers,
                             num_heads=num_heads, hidden_dim=hidden_dim, mlp_dim=mlp_dim, **kwargs)


def vit_s_16(**kwargs: Any) -> VisionTransformer:
    """ViT-S/16 (BASELINE.json configs[1]); not a reference builder, same constructor."""
    return _vision_transformer(16, 12, 6, 384, 1536, **kwargs)


def vit_b_16(**kwargs: Any) -> VisionTransformer:
    return _vision_transformer(16, 12, 12, 768, 3072, **kwargs)       # vit.py:396-403


def vit_b_32(**kwargs: Any) -> VisionTransformer:
    return _vision_transformer(32, 12, 12, 768, 3072, **kwargs)


def vit_l_16(**kwargs: Any) -> VisionTransformer:
    return _vision_transformer(16, 24, 16, 1024, 4096, **kwargs)      # vit.py:454-461


def vit_l_32(**kwargs: Any) -> VisionTransformer:
    return _vision_transformer(32, 24, 16, 1024, 4096, **kwargs)


def vit_h_14(**kwargs: Any) -> VisionTransformer:
    return _vision_transformer(14, 32, 16, 1280, 5120, **kwargs)      # head_dim 80, 257 tokens: the streaming attention kernels (csrc/nrv_attn_gen.hip)
