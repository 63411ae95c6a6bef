"""Which bf16 rounding point carries the deviation of the bf16 path from the fp32 reference?  CPU only.

The full vit_b_16 of the bench (12 layers, 1000 classes, 197 tokens), batch 2, seeds 0..S-1.  For every rounding point of the
HIP path the oracle is evaluated with ONLY that point rounded to bf16, and with every point BUT that one;
deviation = max|logits - fp32 logits| / max|fp32 logits| and |loss - fp32 loss|.  Usage: python tools/parity_attribution.py [seeds] [arch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import simple_vit_oracle as SO  # noqa: E402
from oracle import vit_oracle as VO  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
arch = sys.argv[2] if len(sys.argv) > 2 else "vit_b_16"
CFG = {"vit_b_16": dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072),
       "vit_s_16": dict(image_size=224, patch_size=16, num_layers=12, num_heads=6, hidden_dim=384, mlp_dim=1536)}[arch]
torch.set_num_threads(8)
P = SO._Q.POINTS
rows = {}
for seed in range(seeds):
    sd = VO.vit_init_state_dict(seed=seed, num_classes=1000, **CFG)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (2,), generator=g)
    with torch.no_grad():
        ref = VO.vit_forward(sd, x, patch_size=16, num_heads=CFG["num_heads"])
        lref = SO.cross_entropy_ls(ref, y)

        def dev(points):
            out = VO.vit_forward(sd, x, patch_size=16, num_heads=CFG["num_heads"], emulate_bf16=points)
            return ((out - ref).abs().max() / ref.abs().max()).item(), abs((SO.cross_entropy_ls(out, y) - lref).item())

        cases = [("all points", True)] + [("only " + p, (p,)) for p in P] + [("all but " + p, tuple(q for q in P if q != p)) for p in P]
        cases += [("only operands of the GEMMs (img w xn qkv o h), P fp32", tuple(q for q in P if q != "p")),
                  ("only activations (xn qkv p o h), weights + image fp32", ("xn", "qkv", "p", "o", "h"))]
        for name, pts in cases:
            rows.setdefault(name, []).append(dev(pts))
print(f"{arch}, batch 2, {seeds} seeds: deviation from the fp32 oracle with bf16 rounding at the named points only")
print(f"{'rounding points':62s} {'logits rel (mean / max over seeds)':>36s} {'|loss diff| mean / max':>26s}")
for name, v in rows.items():
    a = [t[0] for t in v]; b = [t[1] for t in v]
    print(f"{name:62s} {sum(a)/len(a):16.2e} / {max(a):9.2e} {sum(b)/len(b):16.2e} / {max(b):9.2e}")
