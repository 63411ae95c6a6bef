"""Masked-autoencoder wrapper on the HIP encoder path (BASELINE.json configs[4]: ViT-B/16 encoder, 75 % mask, 49 tokens).

Same constructor and forward as the reference (`/root/reference/vit_pytorch_robust/mae.py:9-118`): patch embedding of all
tokens, per-sample random permutation, the encoder transformer on the kept 25 %, a narrow decoder transformer on the
re-assembled sequence, MSE on the masked patches.  Unlike the reference file (whose `from vit import Transformer` cannot be
satisfied) it runs: the decoder is `lucid_vit.Transformer`, which is what the reference intends.

Hot parts -- patch unfold + projection (+ positions), token gather / scatter-back, both transformers, `enc_to_dec`,
`to_pixels` -- go through libnrv_hip.so; the index bookkeeping (argsort, mask-token assembly) stays in PyTorch.
`forward(img, rand_indices=None)`: the optional permutation argument exists for reproducible parity tests.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ._lib import PATCH_P1P2C
from .encoder import GatherTokensFn, LinearFn, PatchEmbedFn
from .lucid_vit import Transformer


class MAE(nn.Module):
    def __init__(self, *, encoder, decoder_dim, masking_ratio=0.75, decoder_depth=1, decoder_heads=8, decoder_dim_head=64):
        super().__init__()
        assert 0 < masking_ratio < 1, 'masking ratio must be kept between 0 and 1'
        self.masking_ratio = masking_ratio
        self.encoder = encoder
        num_patches, encoder_dim = encoder.pos_embedding.shape[-2:]
        self.to_patch, self.patch_to_emb = encoder.to_patch_embedding[:2]
        pixel_values_per_patch = self.patch_to_emb.weight.shape[-1]
        self.decoder_dim = decoder_dim
        self.enc_to_dec = nn.Linear(encoder_dim, decoder_dim) if encoder_dim != decoder_dim else nn.Identity()
        self.mask_token = nn.Parameter(torch.randn(decoder_dim))
        self.decoder = Transformer(dim=decoder_dim, depth=decoder_depth, heads=decoder_heads, dim_head=decoder_dim_head,
                                   mlp_dim=decoder_dim * 4)
        self.decoder_pos_emb = nn.Embedding(num_patches, decoder_dim)
        self.to_pixels = nn.Linear(decoder_dim, pixel_values_per_patch)

    def forward(self, img, rand_indices=None):
        device = img.device
        patches = self.to_patch(img).to(torch.float32)                          # [b, n, p*p*c]   (bf16-rounded pixels)
        batch, num_patches, _ = patches.shape
        # patch -> token with the positions of tokens 1..n folded into the GEMM epilogue (mae.py:61-62)
        pos = self.encoder.pos_embedding[:, 1:num_patches + 1]
        tokens = PatchEmbedFn.apply(img, self.patch_to_emb.weight, self.patch_to_emb.bias, pos, None,
                                    self.to_patch.patch_height, PATCH_P1P2C, None)
        num_masked = int(self.masking_ratio * num_patches)
        if rand_indices is None:
            rand_indices = torch.rand(batch, num_patches, device=device).argsort(dim=-1)
        masked_indices, unmasked_indices = rand_indices[:, :num_masked], rand_indices[:, num_masked:]
        batch_range = torch.arange(batch, device=device)[:, None]
        tokens = GatherTokensFn.apply(tokens, unmasked_indices)                 # mae.py:75-76
        masked_patches = patches[batch_range, masked_indices]
        encoded = self.encoder.transformer(tokens)                              # the hot loop at N = n - num_masked
        if isinstance(self.enc_to_dec, nn.Linear):
            dec = LinearFn.apply(encoded, self.enc_to_dec.weight, self.enc_to_dec.bias)
        else:
            dec = encoded
        unmasked_dec = dec + self.decoder_pos_emb(unmasked_indices)
        mask_tokens = self.mask_token[None, None, :].expand(batch, num_masked, -1) + self.decoder_pos_emb(masked_indices)
        full = torch.zeros(batch, num_patches, self.decoder_dim, device=device)
        full[batch_range, unmasked_indices] = unmasked_dec
        full[batch_range, masked_indices] = mask_tokens
        decoded = self.decoder(full)
        pred = LinearFn.apply(decoded[batch_range, masked_indices], self.to_pixels.weight, self.to_pixels.bias)
        return F.mse_loss(pred, masked_patches)
