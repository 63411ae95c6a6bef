#!/usr/bin/env python3
"""Golden fixture for the MAE path, produced by running the reference's mae.py itself (development container only).

mae.py:6 imports `Transformer` from vit.py, which defines none and needs torchvision: as SURVEY.md §8c (5) documents,
the module is importable only if `sys.modules['vit_pytorch_robust.vit']` is pre-seeded with a shim whose `Transformer`
is `learnable_memory_vit.Transformer`; the encoder is `learnable_memory_vit.ViT`.  Fixture metadata says so.
Only data is written (weights, input, the random permutation, loss, a few gradients).
"""
import importlib, os, sys, types
import numpy as np
import torch

REF = "/root/reference/vit_pytorch_robust"
OUT = os.path.dirname(os.path.abspath(__file__))

pkg = types.ModuleType("vit_pytorch_robust"); pkg.__path__ = [REF]
sys.modules["vit_pytorch_robust"] = pkg
lm = importlib.import_module("vit_pytorch_robust.learnable_memory_vit")
shim = types.ModuleType("vit_pytorch_robust.vit"); shim.Transformer = lm.Transformer
sys.modules["vit_pytorch_robust.vit"] = shim
mae_mod = importlib.import_module("vit_pytorch_robust.mae")

torch.manual_seed(0)
enc = lm.ViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256)
mae = mae_mod.MAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=1, decoder_dim_head=64)
g = torch.Generator().manual_seed(99)
img = torch.randn(4, 3, 64, 64, generator=g)
torch.manual_seed(123)
rand_indices = torch.rand(4, 16).argsort(dim=-1)          # what mae.py:67 will draw under the same seed
torch.manual_seed(123)
loss = mae(img)
loss.backward()
out = {"img": img.numpy(), "rand_indices": rand_indices.numpy(), "loss": loss.detach().numpy(),
       "meta": np.array("reference mae.py + shimmed missing import (vit.Transformer := learnable_memory_vit.Transformer)")}
for k, v in mae.state_dict().items():
    out["w." + k] = v.numpy()
for k, p in mae.named_parameters():
    if p.grad is not None:
        out["gn." + k] = p.grad.norm().numpy()
for k in ("encoder.transformer.layers.0.0.to_kv.weight", "decoder.layers.0.1.net.4.weight", "enc_to_dec.weight",
          "encoder.to_patch_embedding.1.weight", "mask_token", "to_pixels.bias"):
    out["g." + k] = dict(mae.named_parameters())[k].grad.numpy()
np.savez_compressed(os.path.join(OUT, "mae_small.npz"), **out)
print("mae_small.npz", os.path.getsize(os.path.join(OUT, "mae_small.npz")), "loss", loss.item())
