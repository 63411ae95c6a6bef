"""Per-half localisation of kernel-vs-oracle differences for the VisionTransformer path (test infrastructure).

For every stage s of the forward (patch embedding, then the attention half and the MLP half of each layer) three numbers
are produced, all max-norm relative to the oracle stream at that stage:

  cum_fp32   HIP stream vs the fp32 oracle stream          (what accumulates from bf16 operands)
  cum_emu    HIP stream vs the bf16-emulating oracle       (accumulated rounding-boundary flips + any kernel error)
  iso_emu    the HIP half applied to the EMULATING ORACLE'S OWN INPUT of that stage vs the emulating oracle's output

`iso_emu` is the number that localises: it removes accumulation, so a rounding point the emulation misses (or a wrong
kernel) shows up as one stage with a large isolated error, while pure rounding-boundary flips give the same small
isolated error at every stage.
"""
from __future__ import annotations

import torch


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def vt_stage_table(model, sd, x_cpu, *, patch_size: int, num_heads: int, dev):
    """Returns [(stage name, cum_fp32, cum_emu, iso_emu)] for a `noise_robust_vit_amd.VisionTransformer` holding `sd`."""
    from noise_robust_vit_amd import encoder as E
    from noise_robust_vit_amd._lib import PATCH_CP1P2
    from oracle import vit_oracle as V

    cap32, cape = {}, {}
    with torch.no_grad():
        V.vit_forward(sd, x_cpu, patch_size=patch_size, num_heads=num_heads, capture=cap32)
        V.vit_forward(sd, x_cpu, patch_size=patch_size, num_heads=num_heads, emulate_bf16=True, capture=cape)
    rows = []
    with torch.no_grad():
        tok = E.PatchEmbedFn.apply(x_cpu.to(dev), model.conv_proj.weight, model.conv_proj.bias, model.encoder.pos_embedding,
                                   model.class_token, model.patch_size, PATCH_CP1P2, None)
        B, S, D = tok.shape
        rows.append(("patch-embed + cls + pos", _rel(tok, cap32["embed"]), _rel(tok, cape["embed"]), _rel(tok, cape["embed"])))
        cur = tok.reshape(B * S, D).contiguous()
        prev_emu = cape["embed"]
        for i, blk in enumerate(model.encoder.layers):
            p = blk.layer_params()
            meta = blk._meta
            for half, key in (("attn", f"layer{i}.attn_out"), ("mlp", f"layer{i}.out")):
                iso_in = prev_emu.to(dev).reshape(B * S, D).contiguous()
                if half == "attn":
                    cur, _ = E.attn_half_fwd(cur, B, S, meta, *p[0:6], residual=True)
                    iso, _ = E.attn_half_fwd(iso_in, B, S, meta, *p[0:6], residual=True)
                else:
                    cur, _ = E.mlp_half_fwd(cur, meta, *p[6:12], residual=True, save=False)
                    iso, _ = E.mlp_half_fwd(iso_in, meta, *p[6:12], residual=True, save=False)
                rows.append((f"layer {i} {half} half", _rel(cur.reshape(B, S, D), cap32[key]), _rel(cur.reshape(B, S, D), cape[key]),
                             _rel(iso.reshape(B, S, D), cape[key])))
                prev_emu = cape[key]
    return rows


def format_table(title: str, rows) -> str:
    out = [title, f"{'stage':28s} {'HIP vs fp32':>12s} {'HIP vs emu':>12s} {'isolated vs emu':>16s}"]
    for name, a, b, c in rows:
        out.append(f"{name:28s} {a:12.3e} {b:12.3e} {c:16.3e}")
    return "\n".join(out)
