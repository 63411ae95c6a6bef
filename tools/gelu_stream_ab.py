#!/usr/bin/env python3
"""What the 8-bit gelu' stream (NRV_EPI_BIAS_GELU_Q8 / NRV_EPI_DGELU_Q8) costs in gradient parity, against the bf16 stream.

(1) BASELINE.json configs[0] on the reference-generated fixture (tests/golden/simplevit_cfg1_*.npz, fp32 reference gradients):
    per-parameter relative L2 of the gradients with either stream.
(2) A ViT-B/16-geometry model (2 layers, batch 8) against the fp32 oracle's gradients, same table.
    python tools/gelu_stream_ab.py            (on the GPU box; the oracle is the checker here, as in tests/)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noise_robust_vit_amd import SimpleViT, encoder          # noqa: E402


def load_npz(path):
    return {k: torch.from_numpy(v) for k, v in np.load(path).items()}


def grads(model, x, y):
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model(x), y, label_smoothing=0.1)
    loss.backward()
    return loss.item(), {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}


def table(title, runs, ref):
    print(f"\n{title}\n{'parameter':58s} {'bf16 stream':>12s} {'8-bit stream':>12s}")
    tot = {k: [0.0, 0.0] for k in runs}
    for name in ref:
        r = ref[name].reshape(-1)
        if r.norm() < 1e-12:
            continue
        row = []
        for k, g in runs.items():
            d = (g[name].reshape(-1) - r)
            row.append((d.norm() / r.norm()).item())
            tot[k][0] += d.norm().item() ** 2; tot[k][1] += r.norm().item() ** 2
        print(f"{name:58s} {row[0]:12.3e} {row[1]:12.3e}")
    print(f"{'all parameters (one vector)':58s} " + " ".join(f"{(v[0] / v[1]) ** 0.5:12.3e}" for v in tot.values()))


def main():
    dev = torch.device("cuda:0")
    gd = os.path.join(ROOT, "tests", "golden")
    sd = load_npz(f"{gd}/simplevit_cfg1_weights.npz")
    g = load_npz(f"{gd}/simplevit_cfg1_softmax.npz")
    model = SimpleViT(image_size=32, patch_size=16, num_classes=100, dim=192, depth=2, heads=3, mlp_dim=768)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    x, y = g["x"].to(dev), g["y"].to(dev)
    runs = {}
    for name, flag in (("bf16", False), ("u8", True)):
        encoder.GELU_STREAM_U8 = flag
        loss, runs[name] = grads(model, x, y)
        print(f"cfg1 {name}: loss {loss:.6f} (fixture {g['loss'].item():.6f})")
    ref = {n: g["grad." + n].float() for n in runs["bf16"]}
    table("configs[0] (SimpleViT dim 192, depth 2, batch 8): gradient rel-L2 from the reference's fp32 gradients", runs, ref)

    # ViT-B/16 geometry, 2 layers, against the fp32 oracle (autograd through oracle/simple_vit_oracle.py on the CPU)
    from oracle import simple_vit_oracle as O
    torch.manual_seed(0)
    model = SimpleViT(image_size=224, patch_size=16, num_classes=100, dim=768, depth=2, heads=12, mlp_dim=3072)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(8, 3, 224, 224); y = torch.randint(0, 100, (8,))
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    logits = O.simple_vit_forward({**sd, **psd}, x, patch_size=16, heads=12)
    torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1).backward()
    ref = {k: v.grad.float() for k, v in psd.items() if v.grad is not None}
    model = model.to(dev).train()
    runs = {}
    for name, flag in (("bf16", False), ("u8", True)):
        encoder.GELU_STREAM_U8 = flag
        _, runs[name] = grads(model, x.to(dev), y.to(dev))
    ref = {n: ref[n] for n in runs["bf16"] if n in ref}
    table("SimpleViT dim 768, 12 heads, mlp 3072, depth 2, 224 px, batch 8: gradient rel-L2 from the fp32 oracle", runs, ref)


if __name__ == "__main__":
    main()
