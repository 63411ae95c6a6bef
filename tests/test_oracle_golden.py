"""CPU: the oracle (oracle/) against the golden fixtures recorded from the reference itself (tests/golden/gen_golden.py).

This is what pins the oracle: if these pass, `oracle.simple_vit_oracle` reproduces
/root/reference/vit_pytorch_robust/simple_vit.py (softmax and Sinkhorn attention), its cross-entropy training loss
(examples/CIFAR100.py:139) and every parameter gradient to fp32 round-off.
"""
import numpy as np
import pytest
import torch

from oracle import simple_vit_oracle as O

TOL = 5e-6      # fp32 CPU vs fp32 CPU: summation order only


def load(path):
    return {k: torch.from_numpy(v) for k, v in np.load(path).items()}


@pytest.mark.parametrize("name,robust", [("softmax", False), ("sinkhorn", True)])
def test_cfg1_logits_loss_grads_and_branch_outputs(golden_dir, name, robust):
    sd = load(f"{golden_dir}/simplevit_cfg1_weights.npz")
    g = load(f"{golden_dir}/simplevit_cfg1_{name}.npz")
    assert sum(v.numel() for v in sd.values()) == 1055524
    logits, loss, grads = O.simple_vit_loss_and_grads(sd, g["x"], g["y"], patch_size=16, heads=3, robust=robust)
    assert (logits - g["logits"]).abs().max() / g["logits"].abs().max() < TOL
    assert abs(loss.item() - g["loss"].item()) < 1e-5
    for k, v in grads.items():
        r = g["grad." + k]
        assert (v - r).norm() / r.norm().clamp_min(1e-20) < 1e-5, k
    cap = {}
    O.simple_vit_forward(sd, g["x"], patch_size=16, heads=3, robust=robust, capture=cap)
    for i in range(2):
        for br in ("attn_branch", "ff_branch"):
            r = g[f"cap.layer{i}.{br}"]
            assert (cap[f"layer{i}.{br}"] - r).abs().max() / r.abs().max() < TOL, (i, br)


def test_224_small_pins_token_grid_and_196_token_attention(golden_dir):
    g = load(f"{golden_dir}/simplevit_224_small.npz")
    sd = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
    for robust, key in ((False, "logits"), (True, "logits_sinkhorn")):
        out = O.simple_vit_forward(sd, g["x"], patch_size=16, heads=1, robust=robust)
        assert (out - g[key]).abs().max() / g[key].abs().max() < TOL


def test_sinkhorn_unit_vector(golden_dir):
    g = load(f"{golden_dir}/sinkhorn_unit.npz")
    out = O.sinkhorn_normalise(torch.softmax(g["scores"], dim=-1))
    assert (out - g["out"]).abs().max() < 1e-6
    # rows sum to one after the final row normalisation (utils.py:1036)
    assert (out.sum(-1) - 1).abs().max() < 1e-5


def test_bf16_emulation_is_close_to_fp32(golden_dir):
    sd = load(f"{golden_dir}/simplevit_cfg1_weights.npz")
    g = load(f"{golden_dir}/simplevit_cfg1_softmax.npz")
    emu = O.simple_vit_forward(sd, g["x"], patch_size=16, heads=3, emulate_bf16=True)
    rel = ((emu - g["logits"]).abs().max() / g["logits"].abs().max()).item()
    assert 1e-5 < rel < 2e-2, rel       # different from fp32 (it rounds) but within the bf16 tolerance


def test_cross_entropy_label_smoothing_matches_torch():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(16, 100, generator=g)
    y = torch.randint(0, 100, (16,), generator=g)
    ref = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1)
    assert abs(O.cross_entropy_ls(logits, y).item() - ref.item()) < 1e-6


def test_mae_oracle_against_shimmed_reference_fixture(golden_dir):
    """mae.py:51-118 with the lucidrains-style encoder; fixture = reference MAE + shimmed missing import."""
    from oracle import mae_oracle as M
    g = np.load(f"{golden_dir}/mae_small.npz")
    assert "shimmed" in str(g["meta"])
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    loss = M.mae_forward(leaves, torch.from_numpy(g["img"]), torch.from_numpy(g["rand_indices"]),
                         patch_size=16, enc_heads=2, dec_heads=1)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    for k in g.files:
        if k.startswith("g."):
            r = torch.from_numpy(g[k])
            assert (leaves[k[2:]].grad - r).norm() / r.norm() < 1e-5, k
