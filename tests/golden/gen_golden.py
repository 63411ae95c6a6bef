#!/usr/bin/env python3
"""Generate golden fixtures by running THE REFERENCE ITSELF (development container only).

Imports ``/root/reference/vit_pytorch_robust/simple_vit.py`` (and ``utils.py``)
through an empty stub package -- the reference's ``__init__.py`` cannot be imported
(missing ``datasets`` submodule, torchvision absent) -- and records
inputs, weights, logits, loss, per-block outputs and all parameter gradients as
``.npz`` data.  Only data is written; no reference source travels.

Run:  python tests/golden/gen_golden.py           (needs /root/reference)
Outputs (committed):
  simplevit_cfg1_weights.npz      seed-0 state_dict of BASELINE.json configs[0]
  simplevit_cfg1_softmax.npz      x, y, logits, loss, branch outputs, grads  (robust=False)
  simplevit_cfg1_sinkhorn.npz     same with robust=True (same weights)
  simplevit_224_small.npz         D=64 L=1 H=1 img224 p16 B=2 : pins 14x14 posemb + N=196 attention
  sinkhorn_unit.npz               utils.SinkhornAttention on a random [2,3,7,7] tensor
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/vit_pytorch_robust"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    pkg = types.ModuleType("vit_pytorch_robust")
    pkg.__path__ = [REF]
    sys.modules["vit_pytorch_robust"] = pkg
    sv = importlib.import_module("vit_pytorch_robust.simple_vit")
    ut = importlib.import_module("vit_pytorch_robust.utils")
    return sv, ut


def run_case(sv, cfg, robust, x, y, state_dict=None, seed=0):
    torch.manual_seed(seed)
    model = sv.SimpleViT(robust=robust, **cfg)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    model.train()
    cap = {}
    hooks = []
    for i, (attn, ff) in enumerate(model.transformer.layers):
        hooks.append(attn.register_forward_hook(
            lambda m, inp, out, i=i: cap.__setitem__(f"layer{i}.attn_branch", out.detach().clone())))
        hooks.append(ff.register_forward_hook(
            lambda m, inp, out, i=i: cap.__setitem__(f"layer{i}.ff_branch", out.detach().clone())))
    logits = model(x)
    # examples/CIFAR100.py:139 / baseline.py:70
    loss = F.cross_entropy(logits, y, label_smoothing=0.1)
    loss.backward()
    for h in hooks:
        h.remove()
    out = {"x": x.numpy(), "y": y.numpy(), "logits": logits.detach().numpy(),
           "loss": loss.detach().numpy()}
    for k, v in cap.items():
        out["cap." + k] = v.numpy()
    for k, p in model.named_parameters():
        out["grad." + k] = p.grad.detach().numpy()
    return model, out


def main():
    sv, ut = import_reference()
    torch.set_num_threads(4)

    # ---- BASELINE.json configs[0]: SimpleViT dim=192 depth=2 heads=3 patch=16 img=32, batch=8
    cfg1 = dict(image_size=32, patch_size=16, num_classes=100, dim=192, depth=2, heads=3, mlp_dim=768)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(8, 3, 32, 32, generator=g)
    y = torch.randint(0, 100, (8,), generator=g)
    model, out = run_case(sv, cfg1, False, x, y)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    np.savez(os.path.join(OUT, "simplevit_cfg1_weights.npz"), **{k: v.numpy() for k, v in sd.items()})
    np.savez(os.path.join(OUT, "simplevit_cfg1_softmax.npz"), **out)
    _, out_r = run_case(sv, cfg1, True, x, y, state_dict=sd)
    np.savez(os.path.join(OUT, "simplevit_cfg1_sinkhorn.npz"), **out_r)

    # ---- reduced 224-px case: 14x14 grid, N=196 tokens, one head
    cfg2 = dict(image_size=224, patch_size=16, num_classes=10, dim=64, depth=1, heads=1, mlp_dim=128)
    g = torch.Generator().manual_seed(4321)
    x2 = torch.randn(2, 3, 224, 224, generator=g)
    y2 = torch.randint(0, 10, (2,), generator=g)
    model2, out2 = run_case(sv, cfg2, False, x2, y2, seed=1)
    out2 = {k: v for k, v in out2.items() if not k.startswith("grad.")}      # keep it small
    # the full-res input is 1.2 MB; store it as float16-exact values so it stays small and exact
    for k, v in model2.state_dict().items():
        out2["w." + k] = v.numpy()
    _, out2r = run_case(sv, cfg2, True, x2, y2, state_dict=model2.state_dict(), seed=1)
    out2["logits_sinkhorn"] = out2r["logits"]
    out2["loss_sinkhorn"] = out2r["loss"]
    np.savez_compressed(os.path.join(OUT, "simplevit_224_small.npz"), **out2)

    # ---- SinkhornAttention unit vector (utils.py:1025-1037)
    g = torch.Generator().manual_seed(7)
    s = torch.randn(2, 3, 7, 7, generator=g) * 2.0
    sk = ut.SinkhornAttention(-1)
    np.savez(os.path.join(OUT, "sinkhorn_unit.npz"), scores=s.numpy(), out=sk(s).numpy())

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
