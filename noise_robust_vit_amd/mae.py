"""Masked-autoencoder wrapper on the HIP encoder path (BASELINE.json configs[4]: ViT-B/16 encoder, 75 % mask, 49 tokens).

Drop-in for the reference's `MAE` (`/root/reference/vit_pytorch_robust/mae.py:9-118`): same constructor keywords, same
parameter names (state_dict contract: `enc_to_dec`, `mask_token`, `decoder.*`, `decoder_pos_emb`, `to_pixels`), same
forward semantics -- embed all patches, keep a random 25 % per sample, run the encoder transformer on the kept tokens,
run a narrow decoder on the re-assembled sequence, mean-squared error on the masked patches.  Unlike the reference file
(whose `from vit import Transformer` cannot be satisfied) it runs: the decoder is `lucid_vit.Transformer`,
which is the module the reference intends.

What goes through libnrv_hip.so: patch unfold + projection with the positions folded into the GEMM epilogue, the token
gather and its scatter-back gradient, both transformers, and the two projections around the decoder.  Index
bookkeeping (argsort, mask-token assembly, the final MSE) stays in PyTorch.
`forward(img, rand_indices=None)`: the optional permutation argument exists for reproducible parity tests.
"""
from __future__ import annotations

import torch
from torch import nn

from ._lib import PATCH_P1P2C
from .encoder import GatherTokensFn, LinearFn, PatchEmbedFn, ScatterTokensFn
from .lucid_vit import Transformer


class MAE(nn.Module):
    def __init__(self, *, encoder, decoder_dim, masking_ratio=0.75, decoder_depth=1, decoder_heads=8, decoder_dim_head=64):
        super().__init__()
        if not 0.0 < masking_ratio < 1.0:
            raise AssertionError("masking ratio must be kept between 0 and 1")
        self.masking_ratio = masking_ratio
        self.encoder = encoder
        self.decoder_dim = decoder_dim
        # what the wrapper needs from the encoder (mae.py:29-31): its unfold + projection pair and the learned positions
        self.to_patch = encoder.to_patch_embedding[0]
        self.patch_to_emb = encoder.to_patch_embedding[1]
        tokens_plus_cls, width = encoder.pos_embedding.shape[1], encoder.pos_embedding.shape[2]
        n_patches = tokens_plus_cls           # the reference sizes its decoder table by pos_embedding.shape[-2] (n + 1)
        pixels = self.patch_to_emb.in_features

        self.enc_to_dec = nn.Identity() if width == decoder_dim else nn.Linear(width, decoder_dim)
        self.mask_token = nn.Parameter(torch.randn(decoder_dim))
        self.decoder = Transformer(dim=decoder_dim, depth=decoder_depth, heads=decoder_heads,
                                   dim_head=decoder_dim_head, mlp_dim=4 * decoder_dim)
        self.decoder_pos_emb = nn.Embedding(n_patches, decoder_dim)
        self.to_pixels = nn.Linear(decoder_dim, pixels)

    def grad_groups(self):
        return self.encoder.transformer.grad_groups() + self.decoder.grad_groups()

    def attach_grad_sink(self, sink) -> None:
        """Both transformers write their weight gradients into the data-parallel runtime's flat buffer directly."""
        self.encoder.transformer.attach_grad_sink(sink)
        self.decoder.attach_grad_sink(sink)

    # -- helpers ----------------------------------------------------------------------------------------------------
    def _embed_all(self, img: torch.Tensor, n: int) -> torch.Tensor:
        """tokens[b, t] = patch_to_emb(patch) + pos_embedding[1 + t]  -- one GEMM, positions in its epilogue (mae.py:61-62)."""
        pos = self.encoder.pos_embedding[:, 1:n + 1]
        return PatchEmbedFn.apply(img, self.patch_to_emb.weight, self.patch_to_emb.bias, pos, None,
                                  self.to_patch.patch_height, PATCH_P1P2C, None)

    def _project(self, layer: nn.Module, x: torch.Tensor) -> torch.Tensor:
        return LinearFn.apply(x, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else x

    def _decoder_input(self, kept_tokens, kept_idx, n):
        """Full-length decoder sequence: projected encoder outputs at the kept positions, the mask token elsewhere, decoder
        positions added to both (mae.py:92-107).  Every position is either kept or masked, so `decoder_pos_emb(kept_idx)` and
        `decoder_pos_emb(masked_idx)` together are the table's first n rows in place: one broadcast add instead of two Embedding
        lookups whose backward sorts indices, and the kept rows go through the bounds-checked scatter kernel instead of index_put."""
        b = kept_tokens.shape[0]
        seq = ScatterTokensFn.apply(kept_tokens, kept_idx, n)                       # zeros at the masked positions
        masked = torch.ones(b, n, 1, device=kept_tokens.device, dtype=seq.dtype)
        masked.scatter_(1, kept_idx.unsqueeze(-1), 0.0)                            # 1 where the mask token goes (no gradient)
        return seq + masked * self.mask_token + self.decoder_pos_emb.weight[:n]

    # -- forward ----------------------------------------------------------------------------------------------------
    def forward(self, img, rand_indices=None):
        # regression targets [b, n, p*p*c] from the fp32 image ('b c (h p1) (w p2) -> b (h w) (p1 p2 c)', mae.py:59): index
        # bookkeeping in PyTorch, NOT the bf16 unfold that feeds the projection GEMM -- the targets are not rounded
        p = self.to_patch.patch_height
        bb, cc, hh, ww = img.shape
        patches = (img.detach().to(torch.float32).reshape(bb, cc, hh // p, p, ww // p, p)
                   .permute(0, 2, 4, 3, 5, 1).reshape(bb, (hh // p) * (ww // p), p * p * cc))
        b, n = patches.shape[0], patches.shape[1]
        n_drop = int(self.masking_ratio * n)
        if rand_indices is None:
            rand_indices = torch.rand(b, n, device=img.device).argsort(dim=-1)      # mae.py:67
        drop_idx, kept_idx = rand_indices[:, :n_drop], rand_indices[:, n_drop:]
        rows = torch.arange(b, device=img.device).unsqueeze(1)

        kept = GatherTokensFn.apply(self._embed_all(img, n), kept_idx)              # [b, n - n_drop, d]
        encoded = self.encoder.transformer(kept)                                    # the hot loop, 49 tokens for ViT-B/16
        decoded = self.decoder(self._decoder_input(self._project(self.enc_to_dec, encoded), kept_idx, n))
        prediction = self._project(self.to_pixels, GatherTokensFn.apply(decoded, drop_idx))
        return torch.nn.functional.mse_loss(prediction, patches[rows, drop_idx])
