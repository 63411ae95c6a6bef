#!/usr/bin/env python3
"""Step time of robust=True (Sinkhorn attention) models at the shapes the fused kernels do not take -- the composed path (nrv_bgemm +
SinkhornAttention on materialised scores, kernels._attn_sinkhorn_*_composed) -- next to the same models with softmax attention
(streaming kernels) and to the fused Sinkhorn kernels at 224 px.  GPU only; dev tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_robust_vit_amd import VisionTransformer, kernels as K

dev = torch.device("cuda:0")
CASES = [("vit_b_16 @ 224 px (197 tokens, fused kernels)", dict(image_size=224, patch_size=16, num_heads=12, hidden_dim=768, mlp_dim=3072), 64),
         ("vit_b_16 @ 384 px (577 tokens)", dict(image_size=384, patch_size=16, num_heads=12, hidden_dim=768, mlp_dim=3072), 32),
         ("vit_h_14 @ 224 px (257 tokens, 16 heads x 80)", dict(image_size=224, patch_size=14, num_heads=16, hidden_dim=1280, mlp_dim=5120), 32)]
LAYERS = 2
for name, cfg, batch in CASES:
    for robust in (False, True):
        torch.manual_seed(0)
        m = VisionTransformer(num_layers=LAYERS, num_classes=10, robust=robust, **cfg).to(dev).train()
        x = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"], device=dev).to(torch.bfloat16)
        y = torch.randint(0, 10, (batch,), device=dev)

        def step():
            for p in m.parameters():
                p.grad = None
            torch.nn.functional.cross_entropy(m(x), y).backward()

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            step()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        with K.LaunchProfile() as prof:
            step()
        summ = prof.summary()
        top = ", ".join(f"{k} {v['ms']:.2f}" for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:5])
        print(f"{name:52s} batch {batch:3d} {LAYERS} layers robust={str(robust):5s}: {ms:8.2f} ms fwd+bwd   [{top}]", flush=True)
        del m
        torch.cuda.empty_cache()
