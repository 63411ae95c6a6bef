"""GPU parity tests, model level: the drop-in modules (HIP path through the C ABI) against
  (1) the golden fixtures recorded from the reference itself (tests/golden, SimpleViT path), and
  (2) the CPU oracle (oracle/, fp32) on seeded inputs, plus its bf16-emulating mode.

Stated tolerances (DESIGN.md "Numerics"):  the hot path feeds bf16 operands to fp32-accumulating MFMAs, so the
north-star "1e-3 relative to the fp32 CPU reference" is not reachable against an fp32 reference (every GEMM output
carries ~1.6e-3 of operand-rounding noise); the ACHIEVED tolerance is the contract and every bound below is <= 2x
what was measured on MI355X for that configuration (max |d| / max |ref| on the logits):
    SimpleViT (mean pooling)         vs fp32 reference <= 6e-3,   vs the bf16-emulating oracle <= 1e-3 (north-star)
    VisionTransformer (class token)  per case in VT_CASES (<= 1.2e-2 vs fp32; 1e-3 ... 8e-3 vs the emulating oracle)
    gradients                        per-parameter relative L2 <= 1e-2 (SimpleViT) / per case (VT), cosine >= 0.999
The comparison with the oracle evaluated with the same bf16 rounding points separates a wrong kernel from operand
rounding; `test_vision_transformer_stage_localisation` shows per stage that what is left is rounding-boundary flips.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL_FP32REF = 6e-3      # SimpleViT, bf16 operands vs fp32 reference, max-norm relative (measured 6.4e-4 ... 2.9e-3)
LOGIT_TOL_EMULATED = 1e-3     # same rounding points as the kernels (north-star tolerance; measured 4.8e-6 ... 6.9e-4)
LOSS_TOL_FP32REF = 3e-3       # measured 5e-6 ... 1.6e-3 (ViT-B geometry)
GRAD_RELL2_TOL = 1e-2         # per-parameter relative L2 vs fp32 autograd of the oracle (measured worst 4.4e-3)
GRAD_COS_TOL = 0.999
BRANCH_TOL_FP32REF = 1.2e-2   # a single half's OUTPUT (no residual stream to dilute it) vs fp32: bf16 operands, measured <= 6e-3


def relmax(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def load_npz(path):
    return {k: torch.from_numpy(v) for k, v in np.load(path).items()}


def check_grads(model, ref_grads, prefix="", tol=None):
    tol = GRAD_RELL2_TOL if tol is None else tol
    worst = (0.0, None)
    for name, p in model.named_parameters():
        assert p.grad is not None, f"no grad for {name}"
        g = p.grad.detach().float().cpu().reshape(-1)
        r = ref_grads[prefix + name].float().reshape(-1)
        if r.norm() < 1e-12:
            assert g.norm() < 1e-6, name
            continue
        rel = ((g - r).norm() / r.norm()).item()
        cos = torch.nn.functional.cosine_similarity(g, r, dim=0).item()
        if rel > worst[0]:
            worst = (rel, name)
        assert rel < tol and cos > GRAD_COS_TOL, f"{name}: rel-L2 {rel:.3e} cos {cos:.6f}"
    return worst


@pytest.mark.parametrize("robust", [False, True])
def test_simplevit_cfg1_against_reference_fixture(dev, golden_dir, robust):
    """BASELINE.json configs[0]: SimpleViT dim=192 depth=2 heads=3 patch=16 img=32 batch=8, reference weights."""
    from noise_robust_vit_amd import SimpleViT
    from oracle import simple_vit_oracle as O
    sd = load_npz(f"{golden_dir}/simplevit_cfg1_weights.npz")
    g = load_npz(f"{golden_dir}/simplevit_cfg1_{'sinkhorn' if robust else 'softmax'}.npz")
    model = SimpleViT(image_size=32, patch_size=16, num_classes=100, dim=192, depth=2, heads=3, mlp_dim=768, robust=robust)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    x, y = g["x"].to(dev), g["y"].to(dev)
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1)      # examples/CIFAR100.py:139
    loss.backward()
    e_ref = relmax(logits, g["logits"])
    emu = O.simple_vit_forward(sd, g["x"], patch_size=16, heads=3, robust=robust, emulate_bf16=True)
    e_emu = relmax(logits, emu)
    print(f"cfg1 robust={robust}: logits vs reference fixture {e_ref:.3e}, vs bf16-emulating oracle {e_emu:.3e}, "
          f"loss {loss.item():.6f} vs {g['loss'].item():.6f}")
    assert e_ref < LOGIT_TOL_FP32REF
    assert e_emu < LOGIT_TOL_EMULATED
    assert abs(loss.item() - g["loss"].item()) < LOSS_TOL_FP32REF
    worst = check_grads(model, g, prefix="grad.")
    print("worst grad rel-L2:", worst)


def test_simplevit_224_small_against_reference_fixture(dev, golden_dir):
    """14x14 token grid, N = 196: pins the sincos table layout and the N=196 attention tiling."""
    from noise_robust_vit_amd import SimpleViT
    from oracle import simple_vit_oracle as O
    g = load_npz(f"{golden_dir}/simplevit_224_small.npz")
    sd = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
    model = SimpleViT(image_size=224, patch_size=16, num_classes=10, dim=64, depth=1, heads=1, mlp_dim=128)
    model.load_state_dict(sd)
    model = model.to(dev)
    logits = model(g["x"].to(dev))
    emu = O.simple_vit_forward(sd, g["x"], patch_size=16, heads=1, emulate_bf16=True)
    e_ref, e_emu = relmax(logits, g["logits"]), relmax(logits, emu)
    print(f"224-small: vs fixture {e_ref:.3e}, vs emulating oracle {e_emu:.3e}")
    assert e_ref < LOGIT_TOL_FP32REF and e_emu < LOGIT_TOL_EMULATED
    # the positional table itself
    assert relmax(model.positional_table(dev), O.posemb_sincos_2d(14, 14, 64)) < 1e-6


def test_standalone_attention_and_feedforward_modules(dev):
    """`Attention(dim, heads, dim_head)(x)` and `FeedForward(dim, hidden)(x)` as separate drop-in modules."""
    from noise_robust_vit_amd import Attention, FeedForward
    from oracle import simple_vit_oracle as O
    torch.manual_seed(0)
    att = Attention(192, heads=3, dim_head=64)
    ff = FeedForward(192, 768)
    x = torch.randn(4, 50, 192)
    sd_a = {"p.norm.weight": att.norm.weight.data, "p.norm.bias": att.norm.bias.data,
            "p.to_qkv.weight": att.to_qkv.weight.data, "p.to_out.weight": att.to_out.weight.data}
    sd_f = {"p." + k: v.data for k, v in ff.state_dict().items()}
    Q = O._Q(False)
    xr = x.clone().requires_grad_(True)
    ref_a = O.attention_block(xr, sd_a, "p.", 3, 64, False, Q)
    ref_f = O.feed_forward_block(xr, sd_f, "p.", Q)
    att, ff = att.to(dev), ff.to(dev)
    xd = x.to(dev).requires_grad_(True)
    out_a, out_f = att(xd), ff(xd)
    print(f"stand-alone Attention {relmax(out_a, ref_a):.3e}, FeedForward {relmax(out_f, ref_f):.3e} vs fp32 oracle")
    assert relmax(out_a, ref_a) < BRANCH_TOL_FP32REF, relmax(out_a, ref_a)
    assert relmax(out_f, ref_f) < BRANCH_TOL_FP32REF, relmax(out_f, ref_f)
    w = torch.randn(4, 50, 192)
    (ref_a * w).sum().backward()
    ga = xr.grad.clone(); xr.grad = None
    (out_a * w.to(dev)).sum().backward()
    rel = ((xd.grad.cpu() - ga).norm() / ga.norm()).item()
    assert rel < GRAD_RELL2_TOL, rel


VT_CASES = {
    # name: (constructor arguments, batch, bound vs fp32 oracle, bound vs bf16-emulating oracle, bound on worst gradient rel-L2)
    # Bounds are <= 2x the values measured on MI355X (DESIGN.md "Numerics" lists the measurements); the north-star 1e-3
    # is met against the emulating oracle only where mean pooling or many tokens average the rounding-boundary flips.
    "d192_l2_n5": (dict(image_size=32, patch_size=16, num_layers=2, num_heads=3, hidden_dim=192, mlp_dim=768, num_classes=10), 5,
                   1.2e-2, 3e-3, 1.8e-2),          # measured 5.9e-3, 1.5e-3, 9.0e-3
    "d128_l1_n17": (dict(image_size=64, patch_size=16, num_layers=1, num_heads=2, hidden_dim=128, mlp_dim=256, num_classes=7), 5,
                    1.1e-2, 1e-3, 1.6e-2),         # measured 5.2e-3, 3.6e-7, 7.9e-3
    # ViT-B/16 geometry (BASELINE.json configs[2], the headline): 12 heads, D 768, M 3072, 197 tokens; 2 of its 12 layers
    "vit_b_16_l2": (dict(image_size=224, patch_size=16, num_layers=2, num_heads=12, hidden_dim=768, mlp_dim=3072, num_classes=13), 2,
                    1.0e-2, 6.5e-3, 1.6e-2),       # measured 4.9e-3, 3.2e-3, 8.0e-3
    # the headline model itself: vit_b_16 as bench.py builds it (12 layers, 1000 classes), batch 2
    "vit_b_16_full": (dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072, num_classes=1000), 2,
                      1.1e-2, 8.5e-3, 1.9e-2),     # measured 5.8e-3, 4.4e-3 (emulating vs fp32 oracle: 5.2e-3), 9.9e-3 (conv_proj.weight); loss 6.6949 vs 6.6962
    # ViT-L/16 geometry (BASELINE.json configs[3]): 16 heads, D 1024, M 4096, 197 tokens; 2 of its 24 layers
    "vit_l_16_l2": (dict(image_size=224, patch_size=16, num_layers=2, num_heads=16, hidden_dim=1024, mlp_dim=4096, num_classes=11), 2,
                    1.0e-2, 1.1e-2, 1.6e-2),       # measured 4.7e-3, 5.5e-3 (the emulating oracle itself is 4.7e-3 from the fp32 one), 7.9e-3
    # the full vit_s_16 of the bench (configs[1]: D 384, 6 heads, M 1536, 12 layers; N = 384 runs the 384 x 128 NT tile), batch 2
    "vit_s_16_full": (dict(image_size=224, patch_size=16, num_layers=12, num_heads=6, hidden_dim=384, mlp_dim=1536, num_classes=1000), 2,
                      1.0e-2, 8.8e-3, 2.2e-2),     # measured 5.3e-3, 4.4e-3 (emulating vs fp32 oracle: 5.2e-3), 1.13e-2 (layer 0 ln_1.weight); loss 7.1418 vs 7.1446
    # vit_h_14 geometry (vit.py:512-519: patch 14, 16 heads x 80, D 1280, M 5120, 257 tokens): 1 of its 32 layers.  Head dim 80 and
    # N > 256 both go through the streaming attention kernels (csrc/nrv_attn_gen.hip)
    "vit_h_14_l1": (dict(image_size=224, patch_size=14, num_layers=1, num_heads=16, hidden_dim=1280, mlp_dim=5120, num_classes=9), 2,
                    1.2e-2, 1.2e-2, 2.0e-2),       # measured 3.2e-3, 2.7e-3 (emulating vs fp32 oracle 4.8e-3), 7.4e-3 (conv_proj.weight)
    # ViT-B/16 at 384 px (577 tokens: what interpolate_embeddings, vit.py:522-603, produces checkpoints for): 1 layer
    "vit_b_16_384px_l1": (dict(image_size=384, patch_size=16, num_layers=1, num_heads=12, hidden_dim=768, mlp_dim=3072, num_classes=9), 1,
                          1.2e-2, 1.2e-2, 2.0e-2),  # measured 5.4e-3, 1.7e-3 (emulating vs fp32 oracle 6.7e-3), 7.5e-3 (conv_proj.weight)
    # the full vit_l_16 of the bench (24 layers, 1000 classes), batch 1
    "vit_l_16_full": (dict(image_size=224, patch_size=16, num_layers=24, num_heads=16, hidden_dim=1024, mlp_dim=4096, num_classes=1000), 1,
                      9.7e-3, 7.2e-3, 2.0e-2),     # measured 4.9e-3, 3.6e-3 (emulating vs fp32 oracle: 4.4e-3), 1.04e-2 (layer 0 ln_1.weight); loss 5.9773 vs 5.9798
}
# loss bound per case where 24 layers of bf16 operand rounding exceed the default (measured 2.5e-3 on vit_l_16_full)
VT_LOSS_BOUNDS = {"vit_l_16_full": 5e-3, "vit_s_16_full": 5.7e-3}      # measured 2.5e-3, 2.9e-3


def _vt_setup(cfg, B, dev):
    from noise_robust_vit_amd import VisionTransformer
    from oracle import vit_oracle as V
    sd = V.vit_init_state_dict(seed=3, **cfg)
    # non-trivial biases / class token so that every epilogue operand is exercised
    g = torch.Generator().manual_seed(5)
    for k in sd:
        if k.endswith("bias") or k == "class_token":
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    model = VisionTransformer(**cfg)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], generator=g)
    y = torch.randint(0, cfg["num_classes"], (B,), generator=g)
    return model, sd, x, y


@pytest.mark.parametrize("case", sorted(VT_CASES))
def test_vision_transformer_against_oracle(dev, case):
    """torchvision-style VisionTransformer (class token, learned positions, biased in/out projections, final LN): logits,
    loss and EVERY parameter gradient against the CPU oracle.  Parity unpinned by the reference (its forward cannot run,
    SURVEY.md §0): the oracle is pinned against torch.nn.MultiheadAttention in tests/test_oracle_vit.py."""
    from oracle import vit_oracle as V
    from oracle.simple_vit_oracle import cross_entropy_ls
    cfg, B, tol_ref, tol_emu, tol_grad = VT_CASES[case]
    model, sd, x, y = _vt_setup(cfg, B, dev)
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1)
    loss.backward()
    torch.set_num_threads(8)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = V.vit_forward(leaves, x, patch_size=cfg["patch_size"], num_heads=cfg["num_heads"])
    ref_loss = cross_entropy_ls(ref, y)
    ref_loss.backward()
    emu = V.vit_forward(sd, x, patch_size=cfg["patch_size"], num_heads=cfg["num_heads"], emulate_bf16=True)
    e_ref, e_emu, e_oo = relmax(logits, ref), relmax(logits, emu), relmax(emu, ref)
    ref_grads = {k: v.grad for k, v in leaves.items()}
    worst = check_grads(model, ref_grads, tol=tol_grad)
    print(f"VT {case}: logits vs fp32 oracle {e_ref:.3e}, vs emulating oracle {e_emu:.3e} (emulating vs fp32 oracle {e_oo:.3e}), "
          f"loss {loss.item():.6f} vs {ref_loss.item():.6f}, worst grad rel-L2 {worst[0]:.3e} ({worst[1]})")
    assert e_ref < tol_ref and e_emu < tol_emu
    assert abs(loss.item() - ref_loss.item()) < VT_LOSS_BOUNDS.get(case, LOSS_TOL_FP32REF)


def _keep_family(seed: int, p: float):
    """site -> reproducible Bernoulli(1 - p) keep mask (uint8) of the asked shape: the same masks for the model and the oracle."""
    def keep(site, shape):
        g = torch.Generator().manual_seed(seed * 1000 + site + 7)
        return (torch.rand(shape, generator=g) >= p).to(torch.uint8)
    return keep


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_vision_transformer_dropout_against_oracle_with_injected_masks(dev, p):
    """`dropout` > 0 in training (vit.py:100-101,112,125,154,175; VERDICT r3 missing #5): the encoder-input, attention-branch,
    GELU-output and MLP-branch dropouts with the SAME keep masks in the model and in the oracle -- logits, loss and every
    parameter gradient; eval mode ignores p."""
    from noise_robust_vit_amd import VisionTransformer
    from oracle import vit_oracle as V
    from oracle.simple_vit_oracle import cross_entropy_ls
    cfg = dict(image_size=64, patch_size=16, num_layers=3, num_heads=3, hidden_dim=192, mlp_dim=768, num_classes=10)
    model, sd, x, y = _vt_setup(dict(cfg), 4, dev)
    model = VisionTransformer(dropout=p, **cfg)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    keep = _keep_family(11, p)
    model.encoder._meta.mask_source = keep
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1)
    loss.backward()
    torch.set_num_threads(8)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = V.vit_forward(leaves, x, patch_size=16, num_heads=3, drop=(p, keep))
    ref_loss = cross_entropy_ls(ref, y)
    ref_loss.backward()
    nodrop = V.vit_forward(sd, x, patch_size=16, num_heads=3)
    e_ref = relmax(logits, ref)
    worst = check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=2.0e-2)
    print(f"VT dropout {p}: logits vs fp32 oracle (same masks) {e_ref:.3e} (the masks move the logits by {relmax(ref, nodrop):.2e}), "
          f"loss {loss.item():.6f} vs {ref_loss.item():.6f}, worst grad rel-L2 {worst[0]:.3e} ({worst[1]})")
    assert e_ref < 1.2e-2 and relmax(ref, nodrop) > 10 * e_ref
    assert abs(loss.item() - ref_loss.item()) < LOSS_TOL_FP32REF
    # without an injected source the masks come from torch's generator: two training forwards differ, eval forwards do not
    model.encoder._meta.mask_source = None
    with torch.no_grad():
        a, b = model(x.to(dev)), model(x.to(dev))
        assert not torch.equal(a, b)
        model.eval()
        c, d = model(x.to(dev)), model(x.to(dev))
        assert torch.equal(c, d) and relmax(c, nodrop) < 1.2e-2


@pytest.mark.parametrize("robust", [False, True])
def test_vision_transformer_attention_dropout_against_oracle_with_injected_masks(dev, robust):
    """`attention_dropout` > 0 (vit.py:108): dropout on the attention weights, composed on the materialised matrix (softmax or
    Sinkhorn), together with `dropout`; the same keep masks in the model and the oracle -- logits, loss, every gradient."""
    from noise_robust_vit_amd import VisionTransformer
    from oracle import vit_oracle as V
    from oracle.simple_vit_oracle import cross_entropy_ls
    cfg = dict(image_size=64, patch_size=16, num_layers=2, num_heads=3, hidden_dim=192, mlp_dim=768, num_classes=10)
    _, sd, x, y = _vt_setup(dict(cfg), 4, dev)
    model = VisionTransformer(dropout=0.1, attention_dropout=0.2, robust=robust, **cfg)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    keep, akeep = _keep_family(21, 0.1), _keep_family(22, 0.2)
    model.encoder._meta.mask_source = lambda site, shape: (akeep if site <= -2 else keep)(site, shape)
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1)
    loss.backward()
    torch.set_num_threads(8)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = V.vit_forward(leaves, x, patch_size=16, num_heads=3, robust=robust, drop=(0.1, keep), attn_drop=(0.2, akeep))
    ref_loss = cross_entropy_ls(ref, y)
    ref_loss.backward()
    only_mlp = V.vit_forward(sd, x, patch_size=16, num_heads=3, robust=robust, drop=(0.1, keep))
    e_ref = relmax(logits, ref)
    worst = check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=2.0e-2)
    print(f"VT attention dropout (robust={robust}): logits vs fp32 oracle (same masks) {e_ref:.3e} (the attention masks move the logits by "
          f"{relmax(ref, only_mlp):.2e}), loss {loss.item():.6f} vs {ref_loss.item():.6f}, worst grad rel-L2 {worst[0]:.3e} ({worst[1]})")
    assert e_ref < 1.2e-2 and relmax(ref, only_mlp) > 5 * e_ref
    assert abs(loss.item() - ref_loss.item()) < LOSS_TOL_FP32REF
    model.eval()
    with torch.no_grad():
        assert relmax(model(x.to(dev)), V.vit_forward(sd, x, patch_size=16, num_heads=3, robust=robust)) < 1.2e-2


def test_vision_transformer_robust_with_dropout_against_oracle(dev):
    """robust=True (Sinkhorn attention) and dropout together: the two options are independent in the module tree (vit.py:98-130)."""
    from noise_robust_vit_amd import VisionTransformer
    from oracle import vit_oracle as V
    cfg = dict(image_size=64, patch_size=16, num_layers=2, num_heads=3, hidden_dim=192, mlp_dim=768, num_classes=10)
    _, sd, x, y = _vt_setup(dict(cfg), 4, dev)
    model = VisionTransformer(dropout=0.2, robust=True, **cfg)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    keep = _keep_family(13, 0.2)
    model.encoder._meta.mask_source = keep
    logits = model(x.to(dev))
    logits.square().mean().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = V.vit_forward(leaves, x, patch_size=16, num_heads=3, robust=True, drop=(0.2, keep))
    ref.square().mean().backward()
    assert relmax(logits, ref) < 1.2e-2
    check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=2.0e-2)


@pytest.mark.parametrize("name,cfg", [
    ("vit_s_16", dict(image_size=224, patch_size=16, num_layers=12, num_heads=6, hidden_dim=384, mlp_dim=1536, num_classes=1000)),
    ("vit_b_16", dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072, num_classes=1000))])
def test_full_model_loss_deviation_is_rounding_noise_not_bias(dev, name, cfg):
    """Loss of the HIP path, of the fp32 oracle and of the bf16-EMULATING oracle (round-to-nearest-even at the kernels' rounding
    points) on the full 12-layer models over three weight / input seeds.  A truncating conversion somewhere in the kernels
    would show as a one-signed offset against the emulation; measured on MI355X (profiles/r02_loss_vs_oracles_full_models.txt):
    HIP - fp32 in [-3.6e-3, +4.4e-3], HIP - emulation in [-1.7e-3, +1.6e-3], both signs for both models."""
    from noise_robust_vit_amd import VisionTransformer
    from oracle import vit_oracle as V
    from oracle.simple_vit_oracle import cross_entropy_ls
    torch.set_num_threads(8)
    rows = []
    for seed in (3, 4, 5):
        sd = V.vit_init_state_dict(seed=seed, **cfg)
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(2, 3, 224, 224, generator=g)
        y = torch.randint(0, 1000, (2,), generator=g)
        m = VisionTransformer(**cfg)
        m.load_state_dict(sd)
        m = m.to(dev).train()
        with torch.no_grad():
            lg = m(x.to(dev)).float().cpu()
            ref = V.vit_forward(sd, x, patch_size=16, num_heads=cfg["num_heads"])
            emu = V.vit_forward(sd, x, patch_size=16, num_heads=cfg["num_heads"], emulate_bf16=True)
        l_hip, l_ref, l_emu = (cross_entropy_ls(t, y).item() for t in (lg, ref, emu))
        rows.append((l_hip - l_ref, l_hip - l_emu, l_emu - l_ref))
        print(f"{name} seed {seed}: loss HIP {l_hip:.6f} fp32 oracle {l_ref:.6f} emulating oracle {l_emu:.6f}  "
              f"HIP-fp32 {rows[-1][0]:+.2e}  HIP-emu {rows[-1][1]:+.2e}  emu-fp32 {rows[-1][2]:+.2e}")
        del m
    assert all(abs(r[0]) < 8e-3 for r in rows), rows          # <= 2x the measured 4.4e-3
    assert all(abs(r[1]) < 3.4e-3 for r in rows), rows        # <= 2x the measured 1.7e-3
    # no one-signed offset against the emulation larger than the emulation's own scatter around the fp32 oracle
    mean_off = sum(r[1] for r in rows) / len(rows)
    assert abs(mean_off) < max(abs(r[2]) for r in rows), (mean_off, rows)


@pytest.mark.parametrize("robust", [False, True])
def test_full_size_properties_batch_256_vit_b_16(dev, robust):
    """BASELINE.json configs[2] at its FULL size (vit_b_16, batch 256: 50 432 token rows per GEMM, too large for the CPU oracle)
    through size-independent properties:
      * determinism: the same forward + backward twice gives bit-identical logits and gradients (no atomics anywhere);
      * sample independence: the logits of samples 0..127 inside the batch of 256 are BIT-equal to the logits of the batch of
        those 128 alone -- every output element's arithmetic (K order of the GEMMs, row statistics, per-head attention) is
        independent of how many rows the launch has, although the two runs use different tile heights and split counts;
      * data-parallel identity: the gradient of the mean loss over 256 samples is the mean of the two half-batch gradients,
        to the bf16 rounding noise of the backward (the half-batch runs see last-bit-different head gradients)."""
    from noise_robust_vit_amd import VisionTransformer
    torch.manual_seed(0)
    cfg = dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072, num_classes=1000)
    model = VisionTransformer(**cfg, robust=robust)          # robust: Sinkhorn attention, one-kernel backward per head
    with torch.no_grad():
        model.heads.head.weight.normal_(0.0, 0.02)           # the reference zero-initialises the head (vit.py:304-306)
    model = model.to(dev).train()
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(256, 3, 224, 224, generator=g, device=dev).to(torch.bfloat16)
    y = torch.randint(0, 1000, (256,), generator=g, device=dev)

    stream_out = []           # the residual stream as the HIP encoder stack leaves it (the final LayerNorm and the head are torch ops)
    run_stack = model.encoder.run_stack

    def recording_stack(t):
        out = run_stack(t)
        stream_out.append(out.detach().clone())
        return out

    model.encoder.run_stack = recording_stack

    def run(xb, yb):
        for p in model.parameters():
            p.grad = None
        stream_out.clear()
        logits = model(xb)
        torch.nn.functional.cross_entropy(logits, yb, label_smoothing=0.1).backward()
        return logits.detach().clone(), stream_out[0], {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    l1, s1, g1 = run(x, y)
    l2, s2, g2 = run(x, y)
    assert torch.equal(l1, l2) and torch.equal(s1, s2)
    assert all(torch.equal(g1[k], g2[k]) for k in g1)
    la, sa, ga = run(x[:128], y[:128])
    lb, sb, gb = run(x[128:], y[128:])
    model.encoder.run_stack = run_stack
    # the HIP path (patch embedding + 12 encoder layers): bit-equal per sample; the classifier head is a torch Linear whose
    # hipBLASLt solution (and with it the summation order) depends on the row count: its logits agree to fp32 rounding
    assert torch.equal(s1[:128], sa) and torch.equal(s1[128:], sb)
    assert relmax(l1[:128], la) < 1e-5 and relmax(l1[128:], lb) < 1e-5
    worst = (0.0, None)
    for k in g1:
        avg = 0.5 * (ga[k] + gb[k])
        rel = ((g1[k] - avg).norm() / g1[k].norm().clamp_min(1e-30)).item()
        worst = max(worst, (rel, k))
        # not fp32-summation-order small: the torch head's logits differ in the last fp32 bits between row counts (above), which
        # flips bf16 roundings of the residual-stream gradient here and there -- the same noise as against the oracle
        assert rel < 2e-2, (k, rel)
    print(f"full size vit_b_16 batch 256 robust={robust}: deterministic, sample-independent (bit-equal); "
          f"grad(256) vs mean of two half-batch grads: worst rel-L2 {worst[0]:.2e} ({worst[1]})")


def test_vision_transformer_robust_vit_b_geometry_against_oracle(dev):
    """robust=True (Sinkhorn attention, utils.py:1031-1037) at the ViT-B/16 geometry -- 12 heads, 197 tokens, 2 layers: logits,
    loss and EVERY parameter gradient against the CPU oracle.  The backward of every head runs the one-kernel Sinkhorn
    backward at NP = 224 (three LDS image slots, transposition chunk of 128 queries)."""
    from noise_robust_vit_amd import VisionTransformer
    from oracle import vit_oracle as V
    from oracle.simple_vit_oracle import cross_entropy_ls
    cfg = dict(image_size=224, patch_size=16, num_layers=2, num_heads=12, hidden_dim=768, mlp_dim=3072, num_classes=13)
    sd = V.vit_init_state_dict(seed=3, **cfg)
    g = torch.Generator().manual_seed(5)
    for k in sd:
        if k.endswith("bias") or k == "class_token":
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    model = VisionTransformer(**cfg, robust=True)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    x = torch.randn(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 13, (2,), generator=g)
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1)
    loss.backward()
    torch.set_num_threads(8)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = V.vit_forward(leaves, x, patch_size=16, num_heads=12, robust=True)
    ref_loss = cross_entropy_ls(ref, y)
    ref_loss.backward()
    emu = V.vit_forward(sd, x, patch_size=16, num_heads=12, robust=True, emulate_bf16=True)
    e_ref, e_emu = relmax(logits, ref), relmax(logits, emu)
    worst = check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=ROBUST_VT_BOUNDS[2])
    print(f"VT robust vit_b_16_l2: logits vs fp32 oracle {e_ref:.3e}, vs emulating oracle {e_emu:.3e}, "
          f"loss {loss.item():.6f} vs {ref_loss.item():.6f}, worst grad rel-L2 {worst[0]:.3e} ({worst[1]})")
    assert e_ref < ROBUST_VT_BOUNDS[0] and e_emu < ROBUST_VT_BOUNDS[1]
    assert abs(loss.item() - ref_loss.item()) < LOSS_TOL_FP32REF


@pytest.mark.parametrize("case", ["vit_h_14_l1", "vit_b_16_384px_l1"])
def test_vision_transformer_robust_beyond_the_fused_shapes_against_oracle(dev, case):
    """robust=True where the head's [N, N] matrix does not stay on chip -- vit_h_14 (257 tokens, 16 heads x 80; vit.py:512-519)
    and ViT-B/16 at 384 px (577 tokens) -- runs the composed path (nrv_bgemm + SinkhornAttention on materialised scores,
    kernels._attn_sinkhorn_*_composed): logits, loss and every parameter gradient against the CPU oracle (round 3: NrvError)."""
    from noise_robust_vit_amd import VisionTransformer
    from oracle import vit_oracle as V
    from oracle.simple_vit_oracle import cross_entropy_ls
    cfg, B, tol_ref, tol_emu, tol_grad = VT_CASES[case]
    sd = V.vit_init_state_dict(seed=3, **cfg)
    g = torch.Generator().manual_seed(5)
    for k in sd:
        if k.endswith("bias") or k == "class_token":
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    model = VisionTransformer(**cfg, robust=True)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], generator=g)
    y = torch.randint(0, cfg["num_classes"], (B,), generator=g)
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1)
    loss.backward()
    torch.set_num_threads(8)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = V.vit_forward(leaves, x, patch_size=cfg["patch_size"], num_heads=cfg["num_heads"], robust=True)
    ref_loss = cross_entropy_ls(ref, y)
    ref_loss.backward()
    emu = V.vit_forward(sd, x, patch_size=cfg["patch_size"], num_heads=cfg["num_heads"], robust=True, emulate_bf16=True)
    e_ref, e_emu = relmax(logits, ref), relmax(logits, emu)
    worst = check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=tol_grad)
    print(f"VT robust {case}: logits vs fp32 oracle {e_ref:.3e}, vs emulating oracle {e_emu:.3e}, "
          f"loss {loss.item():.6f} vs {ref_loss.item():.6f}, worst grad rel-L2 {worst[0]:.3e} ({worst[1]})")
    assert e_ref < tol_ref and e_emu < tol_emu
    assert abs(loss.item() - ref_loss.item()) < LOSS_TOL_FP32REF


def test_simplevit_robust_with_other_head_dim(dev):
    """SimpleViT(dim_head=32, robust=True) (simple_vit.py:56-57,101-114): Sinkhorn attention at a head dim the fused kernel does
    not take -- the composed path, against the oracle with all gradients."""
    from noise_robust_vit_amd import SimpleViT
    from oracle import simple_vit_oracle as O
    torch.manual_seed(0)
    model = SimpleViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=3, mlp_dim=256, dim_head=32, robust=True)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    y = torch.randint(0, 10, (3,), generator=torch.Generator().manual_seed(4))
    logits = model(x.to(dev))
    torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1).backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.simple_vit_forward(leaves, x, patch_size=16, heads=3, dim_head=32, robust=True)
    O.cross_entropy_ls(ref, y).backward()
    emu = O.simple_vit_forward(sd, x, patch_size=16, heads=3, dim_head=32, robust=True, emulate_bf16=True)
    assert relmax(logits, ref) < LOGIT_TOL_FP32REF, relmax(logits, ref)
    assert relmax(logits, emu) < LOGIT_TOL_EMULATED, relmax(logits, emu)
    check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=GRAD_RELL2_TOL)


ROBUST_VT_BOUNDS = (1.0e-2, 6.4e-3, 1.5e-2)      # measured on MI355X: 5.0e-3, 3.2e-3, 7.7e-3 (conv_proj.weight); loss 2.48822 vs 2.48793


@pytest.mark.parametrize("case", ["d192_l2_n5", "vit_b_16_l2"])
def test_vision_transformer_stage_localisation(dev, case):
    """Where do kernel and bf16-emulating oracle part ways?  Per stage (patch embedding, every attention / MLP half):
    the accumulated difference and the ISOLATED difference (the HIP half applied to the emulating oracle's own input of
    that stage).  A rounding point the emulation misses, or a wrong kernel, is one stage with a large isolated error;
    rounding-boundary flips give every stage the same small one.  The table is printed and, on the GPU box, written to
    gpurun_out/ (profiles/r02_vt_stage_parity.txt is a committed copy)."""
    import os
    from parity_tools import format_table, vt_stage_table
    cfg, B, _, _, _ = VT_CASES[case]
    model, sd, x, _ = _vt_setup(cfg, B, dev)
    rows = vt_stage_table(model, sd, x, patch_size=cfg["patch_size"], num_heads=cfg["num_heads"], dev=dev)
    text = format_table(f"VisionTransformer {case} (batch {B}): max-norm relative differences of the residual stream", rows)
    print("\n" + text)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"vt_stage_parity_{case}.txt"), "w") as f:
            f.write(text + "\n")
    except OSError:
        pass
    iso = sorted(r[3] for r in rows)
    # Isolated, a half on identical inputs equals the emulation to fp32 rounding (~1e-7) -- unless one of its bf16 stores
    # lands on a rounding boundary and flips (fp32 summation order differs), which shows as ~1e-4 on that stage only.
    # A rounding point missed by the emulation (or a wrong kernel) would make EVERY stage of that kind large: so the
    # median must be at fp32-rounding level and no stage may exceed the north-star 1e-3.
    assert iso[-1] < 1e-3, text
    if cfg["image_size"] < 224:      # 5 tokens: most stages see no flip at all (at 197 tokens x D 768 every stage has a few hundred)
        assert iso[len(iso) // 2] < 1e-5, text


def test_simplevit_s16_depth12_against_oracle(dev):
    """ViT-S/16 geometry (BASELINE.json configs[1]) at batch 2: 12 layers of accumulated bf16 rounding, logits, loss and
    every parameter gradient."""
    from noise_robust_vit_amd import SimpleViT
    from oracle import simple_vit_oracle as O
    torch.manual_seed(0)
    model = SimpleViT(image_size=224, patch_size=16, num_classes=1000, dim=384, depth=12, heads=6, mlp_dim=1536)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (2,), generator=g)
    model = model.to(dev).train()
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1)
    loss.backward()
    torch.set_num_threads(8)
    ref, ref_loss, ref_grads = O.simple_vit_loss_and_grads(sd, x, y, patch_size=16, heads=6)
    emu = O.simple_vit_forward(sd, x, patch_size=16, heads=6, emulate_bf16=True)
    e_ref, e_emu = relmax(logits, ref), relmax(logits, emu)
    worst = check_grads(model, ref_grads)          # measured 4.9e-3
    print(f"SimpleViT-S/16 depth 12: logits vs fp32 oracle {e_ref:.3e}, vs emulating oracle {e_emu:.3e}, "
          f"loss {loss.item():.6f} vs {ref_loss.item():.6f}, worst grad rel-L2 {worst[0]:.3e} ({worst[1]})")
    assert e_ref < LOGIT_TOL_FP32REF
    assert e_emu < LOGIT_TOL_EMULATED            # measured 3.1e-4: mean pooling averages the rounding-boundary flips
    assert abs(loss.item() - ref_loss.item()) < LOSS_TOL_FP32REF


def test_simplevit_other_input_size_than_constructed(dev):
    """The reference rebuilds the sincos table from the actual patch grid on every forward (simple_vit.py:141-143), so a
    SimpleViT accepts any image size divisible by the patch: here a model constructed for 32 x 32 runs 64 x 48 (a 4 x 3
    grid, non-square) and must match the oracle; the table is NOT read past its end."""
    from noise_robust_vit_amd import SimpleViT
    from oracle import simple_vit_oracle as O
    torch.manual_seed(0)
    model = SimpleViT(image_size=32, patch_size=16, num_classes=10, dim=128, depth=1, heads=2, mlp_dim=256)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    g = torch.Generator().manual_seed(3)
    for hw in ((64, 48), (32, 32), (16, 80)):
        x = torch.randn(3, 3, *hw, generator=g)
        logits = model(x.to(dev))
        emu = O.simple_vit_forward(sd, x, patch_size=16, heads=2, emulate_bf16=True)
        assert relmax(logits, emu) < LOGIT_TOL_EMULATED, (hw, relmax(logits, emu))


def test_checkpoint_at_higher_resolution_through_interpolate_embeddings(dev):
    """What interpolate_embeddings exists for (vit.py:522-603): a checkpoint trained at 224 px (197 tokens) is resized to
    384 px (577 tokens) and runs there -- the attention goes through the streaming kernels (N > 256).  Checked against the
    oracle run on the SAME interpolated state dict."""
    from collections import OrderedDict
    from noise_robust_vit_amd.vit import VisionTransformer, interpolate_embeddings
    from oracle import vit_oracle as V
    cfg = dict(patch_size=16, num_layers=1, num_heads=4, hidden_dim=256, mlp_dim=512, num_classes=7)
    torch.manual_seed(0)
    src = VisionTransformer(image_size=224, **cfg)
    torch.nn.init.normal_(src.heads.head.weight, std=0.05)
    sd224 = OrderedDict((k, v.detach().clone()) for k, v in src.state_dict().items())
    sd384 = interpolate_embeddings(384, 16, sd224)
    assert sd384["encoder.pos_embedding"].shape == (1, 577, 256)
    model = VisionTransformer(image_size=384, **cfg)
    model.load_state_dict(sd384)
    model = model.to(dev).eval()
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        logits = model(x.to(dev))
    sd = {k: v.detach().clone() for k, v in sd384.items()}
    ref = V.vit_forward(sd, x, patch_size=16, num_heads=4)
    emu = V.vit_forward(sd, x, patch_size=16, num_heads=4, emulate_bf16=True)
    e_ref, e_emu = relmax(logits, ref), relmax(logits, emu)
    print(f"224 -> 384 px checkpoint (577 tokens): logits vs fp32 oracle {e_ref:.3e}, vs emulating oracle {e_emu:.3e}")
    assert e_ref < 1.2e-2 and e_emu < 1.2e-2


def test_simplevit_with_other_head_dim(dev):
    """SimpleViT(dim_head=32) (simple_vit.py:101-114: inner = heads * dim_head != dim): the streaming attention kernels take
    head dims 32 / 80 / 96 / 128 beside the single-pass kernels' 64."""
    from noise_robust_vit_amd import SimpleViT
    from oracle import simple_vit_oracle as O
    for dim_head, heads in ((32, 3), (96, 2), (128, 1)):
        torch.manual_seed(0)
        model = SimpleViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=heads, mlp_dim=256, dim_head=dim_head)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model = model.to(dev)
        x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(3))
        y = torch.randint(0, 10, (3,), generator=torch.Generator().manual_seed(4))
        logits = model(x.to(dev))
        torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.1).backward()
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = O.simple_vit_forward(leaves, x, patch_size=16, heads=heads, dim_head=dim_head)
        O.cross_entropy_ls(ref, y).backward()
        emu = O.simple_vit_forward(sd, x, patch_size=16, heads=heads, dim_head=dim_head, emulate_bf16=True)
        assert relmax(logits, ref) < LOGIT_TOL_FP32REF, (dim_head, relmax(logits, ref))
        assert relmax(logits, emu) < LOGIT_TOL_EMULATED, (dim_head, relmax(logits, emu))
        check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=GRAD_RELL2_TOL)


def test_lucid_vit_rejects_other_input_size(dev):
    """lucidrains-style ViT has a learned [1, n + 1, D] table: another grid is a shape error in the reference
    (learnable_memory_vit.py: x += self.pos_embedding[:, :(n + 1)] cannot broadcast) and must be one here, not an
    out-of-bounds read of the table."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from noise_robust_vit_amd._lib import NrvError
    vit = ViT(image_size=32, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=128).to(dev)
    with pytest.raises((NrvError, RuntimeError)):
        vit(torch.randn(2, 3, 64, 64, device=dev))


@pytest.mark.parametrize("batch_first", [True, False])
def test_multihead_attention_forward_matches_torch_module(dev, batch_first):
    """`MultiheadAttention.forward(x, x, x, need_weights=False) -> (out, None)` (reference: the forked module at
    utils.py:650-751, whose own forward raises upstream) against installed torch.nn.MultiheadAttention with the same
    packed parameters: output and every gradient, on the CPU in fp32."""
    from noise_robust_vit_amd.vit import MultiheadAttention
    torch.manual_seed(0)
    E, H, B, S = 192, 3, 4, 37
    mine = MultiheadAttention(E, H, batch_first=batch_first)
    with torch.no_grad():
        mine.in_proj_bias.normal_(std=0.1); mine.out_proj.bias.normal_(std=0.1)
    ref = torch.nn.MultiheadAttention(E, H, batch_first=batch_first)
    ref.load_state_dict({k: v.detach().clone() for k, v in mine.state_dict().items()})
    x = torch.randn(B, S, E) if batch_first else torch.randn(S, B, E)
    w = torch.randn_like(x)
    xr = x.clone().requires_grad_(True)
    out_r, _ = ref(xr, xr, xr, need_weights=False)
    (out_r * w).sum().backward()
    mine = mine.to(dev)
    xd = x.to(dev).requires_grad_(True)
    out, attn_w = mine(xd, xd, xd, need_weights=False)
    assert attn_w is None and out.shape == x.shape
    (out * w.to(dev)).sum().backward()
    e = relmax(out, out_r)
    print(f"MultiheadAttention.forward batch_first={batch_first}: {e:.3e} vs torch.nn.MultiheadAttention (fp32)")
    assert e < BRANCH_TOL_FP32REF
    pairs = [(xd.grad, xr.grad, "x")] + [(p.grad, dict(ref.named_parameters())[n].grad, n) for n, p in mine.named_parameters()]
    for g, r, name in pairs:
        g = g.detach().float().cpu().reshape(-1); r = r.reshape(-1)
        rel = ((g - r).norm() / r.norm()).item()
        assert rel < 2e-2, (name, rel)
    # need_weights=True: the attention weights torch's module returns (averaged over the heads by default), recomputed from q, k
    # and the saved log-sum-exp; the output is the same fused result
    with torch.no_grad():
        out2, w_avg = mine(xd, xd, xd, need_weights=True)
        _, w_all = mine(xd, xd, xd, need_weights=True, average_attn_weights=False)
        _, rw_avg = ref(x, x, x, need_weights=True)
        _, rw_all = ref(x, x, x, need_weights=True, average_attn_weights=False)
    assert torch.equal(out2, out.detach())
    assert w_avg.shape == rw_avg.shape == (B, S, S) and w_all.shape == rw_all.shape == (B, H, S, S)
    assert (w_avg.cpu() - rw_avg).abs().max().item() < 5e-3 and (w_all.cpu() - rw_all).abs().max().item() < 5e-3
    with pytest.raises(NotImplementedError):                 # the weights of a masked call are not recomputed (composed path)
        mine(xd, xd, xd, attn_mask=torch.zeros(S, S, device=dev), need_weights=True)
    with torch.no_grad():                                    # a zero additive mask is the unmasked attention, on the composed path
        out3, _ = mine(xd, xd, xd, attn_mask=torch.zeros(S, S, device=dev))
    assert relmax(out3, out.detach()) < 1.2e-2


def test_state_dict_roundtrip_and_legacy_mlp_keys(dev):
    from noise_robust_vit_amd import VisionTransformer
    m = VisionTransformer(image_size=32, patch_size=16, num_layers=1, num_heads=1, hidden_dim=64, mlp_dim=128, num_classes=3)
    sd = m.state_dict()
    legacy = {}
    for k, v in sd.items():
        k2 = k.replace(".mlp.0.", ".mlp.linear_1.").replace(".mlp.3.", ".mlp.linear_2.")
        legacy[k2] = v.clone() + 1.0
    # a version-1 checkpoint (no metadata) with linear_1/linear_2 names must load (vit.py:55-84)
    m.load_state_dict(legacy)
    assert torch.equal(m.encoder.layers.encoder_layer_0.mlp[0].weight.data, legacy["encoder.layers.encoder_layer_0.mlp.linear_1.weight"])


def test_mae_against_shimmed_reference_fixture(dev, golden_dir):
    """BASELINE.json configs[4] path at reduced size: MAE over a lucidrains-style ViT, 75 % mask.
    Fixture = the reference's mae.py run with the shim of SURVEY.md §8c (5) ("reference MAE + shimmed missing import")."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from noise_robust_vit_amd.mae import MAE
    g = load_npz(f"{golden_dir}/mae_small.npz") if False else dict(np.load(f"{golden_dir}/mae_small.npz"))
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w.")}
    enc = ViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256)
    mae = MAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=1, decoder_dim_head=64)
    assert sorted(mae.state_dict()) == sorted(sd)
    mae.load_state_dict(sd)
    mae = mae.to(dev).train()
    loss = mae(torch.from_numpy(g["img"]).to(dev), rand_indices=torch.from_numpy(g["rand_indices"]).to(dev))
    loss.backward()
    ref = float(g["loss"])
    print(f"MAE loss {loss.item():.6f} vs reference {ref:.6f}")
    assert abs(loss.item() - ref) < 5e-3 * abs(ref)
    params = dict(mae.named_parameters())
    for k, v in g.items():
        if k.startswith("g."):
            r = torch.from_numpy(v).reshape(-1)
            gk = params[k[2:]].grad.detach().float().cpu().reshape(-1)
            rel = ((gk - r).norm() / r.norm()).item()
            assert rel < GRAD_RELL2_TOL, (k, rel)
        if k.startswith("gn."):
            assert params[k[3:]].grad is not None, k


@pytest.mark.parametrize("robust", [False, True])
def test_recorder_attention_maps(dev, robust):
    """Recorder(vit)(img) -> (preds, attns [B, depth, heads, N, N]) as in the reference's recorder.py; the maps are the
    (Sinkhorn-normalised for robust=True) softmax of the layer's own q, k: rows sum to 1, and they reproduce the layer's
    attention output when applied to v."""
    from noise_robust_vit_amd import SimpleViT
    from noise_robust_vit_amd import encoder
    from noise_robust_vit_amd.recorder import Recorder
    torch.manual_seed(0)
    vit = SimpleViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, robust=robust).to(dev).eval()
    x = torch.randn(3, 3, 64, 64, device=dev)
    rec = Recorder(vit)
    with torch.no_grad():
        preds, attns = rec(x)
        plain = vit(x)
    assert torch.equal(preds, plain)
    assert attns.shape == (3, 2, 2, 16, 16) and attns.dtype == torch.float32
    assert (attns.sum(-1) - 1).abs().max().item() < 1e-3
    if robust:
        assert (attns.sum(-2) - 1).abs().max().item() < 5e-2          # columns are close to 1 after 3 Sinkhorn rounds
    assert rec.eject() is vit
    with pytest.raises(AssertionError):
        rec(x)
    assert encoder._RECORDING is None


def test_inference_path_equals_training_forward(dev):
    """Under torch.no_grad the encoder keeps nothing for a backward (no gelu' stream, no saved activations): the logits must
    be bit-identical to the training-mode forward."""
    from noise_robust_vit_amd import SimpleViT
    torch.manual_seed(0)
    vit = SimpleViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256).to(dev)
    x = torch.randn(4, 3, 64, 64, device=dev)
    with torch.no_grad():
        a = vit(x)
    b = vit(x)
    assert b.requires_grad and not a.requires_grad
    assert torch.equal(a, b.detach())
    b.sum().backward()
    assert vit.transformer.layers[0][1].net[1].weight.grad is not None


def test_training_actually_learns(dev):
    """End-to-end: Trainer (HIP forward / backward, fused clip + AdamW, batched weight re-staging) overfits one fixed batch.
    Guards the whole loop -- e.g. the bf16 weight images going stale after an optimizer step, which no single-step parity
    test can see."""
    from noise_robust_vit_amd import SimpleViT
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    torch.manual_seed(0)
    vit = SimpleViT(image_size=32, patch_size=8, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256).to(dev).train()
    tr = Trainer(vit, TrainConfig(lr=2e-3, weight_decay=0.05, grad_max_norm=5.0, label_smoothing=0.0))
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(32, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, 10, (32,), generator=g).to(dev)
    first = tr.step(x, y).item()
    for _ in range(60):
        last = tr.step(x, y).item()
    assert first > 2.0 and last < 0.25 * first, (first, last)
    vit.eval()
    with torch.no_grad():
        acc = (vit(x).argmax(-1) == y).float().mean().item()
    assert acc > 0.9, acc


def test_mae_vit_b_geometry_49_tokens_against_oracle(dev):
    """BASELINE.json configs[4] at its real geometry: ViT-B/16 encoder width (D 768, 12 heads, M 3072) on the 49 kept of 196
    tokens (2 of its 12 layers), the wrapper's 512-wide decoder on all 196; loss and every gradient against the CPU
    oracle (oracle/mae_oracle.py, itself pinned by the reference-generated fixture above)."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from noise_robust_vit_amd.mae import MAE
    from oracle import mae_oracle as MO
    torch.manual_seed(0)
    enc = ViT(image_size=224, patch_size=16, num_classes=10, dim=768, depth=2, heads=12, mlp_dim=3072)
    mae = MAE(encoder=enc, decoder_dim=512, masking_ratio=0.75, decoder_depth=1, decoder_heads=8, decoder_dim_head=64)
    with torch.no_grad():                                  # the reference initialises these N(0, 1): keep the tokens O(1)
        enc.pos_embedding.mul_(0.02); enc.cls_token.mul_(0.02)
    sd = {k: v.detach().clone() for k, v in mae.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    img = torch.randn(2, 3, 224, 224, generator=g)
    idx = torch.rand(2, 196, generator=g).argsort(dim=-1)
    mae = mae.to(dev).train()
    loss = mae(img.to(dev), rand_indices=idx.to(dev))
    loss.backward()
    torch.set_num_threads(8)
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = MO.mae_forward(leaves, img, idx, patch_size=16, enc_heads=12, dec_heads=8)
    ref.backward()
    print(f"MAE ViT-B geometry: loss {loss.item():.6f} vs oracle {ref.item():.6f}")
    assert abs(loss.item() - ref.item()) < 2e-3 * abs(ref.item())
    worst = (0.0, None)
    for k, p in mae.named_parameters():
        r = leaves[k].grad
        if r is None:                                       # cls_token / mlp_head are not on the MAE path (mae.py:84)
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
            continue
        gk = p.grad.detach().float().cpu().reshape(-1); r = r.reshape(-1)
        rel = ((gk - r).norm() / r.norm().clamp_min(1e-30)).item()
        worst = max(worst, (rel, k))
        assert rel < 1.8e-2, (k, rel)                    # measured worst 8.7e-3
    print("worst grad rel-L2:", worst)


@pytest.mark.parametrize("kind", ["causal_bool", "float_per_head", "key_padding", "both"])
def test_multihead_attention_masks_against_torch_module(dev, kind):
    """`attn_mask` / `key_padding_mask` of the stand-alone MultiheadAttention (utils.py:741-751: the forked torch module's forward
    signature) on the composed path, against torch.nn.MultiheadAttention itself (CPU fp32) with the same weights: output and the
    gradients of the input and of every parameter."""
    from noise_robust_vit_amd.vit import MultiheadAttention
    torch.manual_seed(0)
    B, N, E, H = 3, 10, 64, 2
    ref = torch.nn.MultiheadAttention(E, H, batch_first=True)
    with torch.no_grad():
        ref.in_proj_bias.normal_(std=0.1); ref.out_proj.bias.normal_(std=0.1)
    ours = MultiheadAttention(E, H, batch_first=True)
    ours.load_state_dict(ref.state_dict())
    ours = ours.to(dev)
    x = torch.randn(B, N, E)
    am = kp = None
    if kind in ("causal_bool", "both"):
        am = torch.triu(torch.ones(N, N, dtype=torch.bool), diagonal=1)
    if kind == "float_per_head":
        am = torch.randn(B * H, N, N) * 0.5
    if kind in ("key_padding", "both"):
        kp = torch.zeros(B, N, dtype=torch.bool); kp[0, 7:] = True; kp[2, 9:] = True
    xr = x.clone().requires_grad_(True)
    yr, _ = ref(xr, xr, xr, attn_mask=am, key_padding_mask=kp, need_weights=False)
    w = torch.randn(B, N, E)
    (yr * w).sum().backward()
    xo = x.to(dev).requires_grad_(True)
    yo, none = ours(xo, xo, xo, attn_mask=am, key_padding_mask=kp, need_weights=False)
    assert none is None
    (yo * w.to(dev)).sum().backward()
    assert relmax(yo, yr) < 1.2e-2, relmax(yo, yr)
    assert ((xo.grad.cpu() - xr.grad).norm() / xr.grad.norm()).item() < 2e-2
    for (k, po), (_, pr) in zip(ours.named_parameters(), ref.named_parameters()):
        rel = ((po.grad.cpu() - pr.grad).norm() / pr.grad.norm().clamp_min(1e-30)).item()
        assert rel < 2e-2, (k, rel)


def test_lucid_vit_dropout_against_oracle_with_injected_masks(dev):
    """lucidrains-style ViT (the MAE encoder family) with `dropout` and `emb_dropout` > 0 (learnable_memory_vit.py:37,39,54,61,83,126,142):
    the one dropout probability at its four sites per layer -- attention weights (composed path), to_out, GELU output, second Linear --
    plus the embedding dropout; same keep masks in the oracle's restatement: logits and every gradient."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from oracle import mae_oracle as MO
    from oracle.simple_vit_oracle import layer_norm, patchify_p1p2c, _Q
    torch.manual_seed(0)
    cfg = dict(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256)
    model = ViT(dropout=0.15, emb_dropout=0.1, **cfg)
    with torch.no_grad():
        model.pos_embedding.mul_(0.02); model.cls_token.mul_(0.02)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 3, 64, 64, generator=g)
    keep, ekeep = _keep_family(31, 0.15), _keep_family(32, 0.1)
    model = model.to(dev).train()
    model.transformer._meta.mask_source = lambda site, shape: (ekeep if site == -1 else keep)(site, shape)
    logits = model(x.to(dev))
    logits.square().mean().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}

    def oracle(drop, edrop):
        t = patchify_p1p2c(x, 16, 16) @ leaves["to_patch_embedding.1.weight"].t() + leaves["to_patch_embedding.1.bias"]
        t = torch.cat([leaves["cls_token"].expand(4, -1, -1), t], dim=1) + leaves["pos_embedding"]
        t = MO._drop(t, edrop, -1)
        t = MO.lucid_transformer(t, leaves, "transformer.", 2, 64, _Q(False), drop)
        h = layer_norm(t[:, 0], leaves["mlp_head.0.weight"], leaves["mlp_head.0.bias"], 1e-5)
        return h @ leaves["mlp_head.1.weight"].t() + leaves["mlp_head.1.bias"]

    ref = oracle((0.15, keep), (0.1, ekeep))
    ref.square().mean().backward()
    with torch.no_grad():
        plain = oracle(None, None)
    e = relmax(logits, ref)
    print(f"lucid ViT dropout: logits vs fp32 oracle (same masks) {e:.3e}; the masks move the logits by {relmax(ref, plain):.2e}")
    assert e < 1.2e-2 and relmax(ref, plain) > 10 * e
    check_grads(model, {k: v.grad for k, v in leaves.items()}, tol=2.0e-2)


def test_lucid_standalone_attention_and_feedforward_dropout_against_oracle(dev):
    """`Attention(dropout=p)` / `FeedForward(dropout=p)` used on their own (learnable_memory_vit.py:30-86): the dropout on the attention
    weights and behind to_out, behind the GELU and behind the second Linear, with the oracle's masks; eval mode ignores p."""
    from noise_robust_vit_amd.lucid_vit import Attention, FeedForward
    from oracle import mae_oracle as MO
    from oracle.simple_vit_oracle import _Q
    torch.manual_seed(3)
    p = 0.2
    keep = _keep_family(41, p)
    x = torch.randn(3, 50, 128)
    for kind in ("attention", "feed_forward"):
        mod = Attention(128, heads=2, dim_head=64, dropout=p) if kind == "attention" else FeedForward(128, 256, dropout=p)
        sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}
        mod = mod.to(dev).train()
        mod._meta.mask_source = keep
        xin = x.to(dev).requires_grad_(True)
        y = mod(xin)
        y.square().mean().backward()
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        f = (lambda t, d: MO.lucid_attention(t, leaves, "", 2, 64, _Q(False), d, 0)) if kind == "attention" else \
            (lambda t, d: MO.lucid_feed_forward(t, leaves, "", _Q(False), d, 0))
        ref = f(xr, (p, keep))
        ref.square().mean().backward()
        with torch.no_grad():
            plain = f(x, None)
        e = relmax(y, ref)
        print(f"stand-alone {kind} with dropout {p}: vs fp32 oracle (same masks) {e:.3e}; the masks move the output by {relmax(ref, plain):.2e}")
        assert e < 1.2e-2 and relmax(ref, plain) > 10 * e
        assert ((xin.grad.cpu() - xr.grad).norm() / xr.grad.norm()).item() < 2e-2
        check_grads(mod, {k: v.grad for k, v in leaves.items()}, tol=2.0e-2)
        mod.eval()
        with torch.no_grad():
            assert relmax(mod(x.to(dev)), plain) < 1.2e-2


def test_mae_gradients_through_the_reducer_sink_match_autograd(dev):
    """With a GradReducer attached (every training step: train.Trainer builds one even on a single GPU) both MAE transformers
    write their weight gradients into the flat buffer directly -- the fused [to_q; to_kv] projection as one TN GEMM per
    parameter on its column block of dQKV.  Same loss and the same gradients as the plain autograd run (bit-equal up to the
    summation order of the split-K slabs), every parameter with a gradient reported as ready, and a second step overwrites
    rather than accumulates."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from noise_robust_vit_amd.mae import MAE
    from noise_robust_vit_amd.parallel import GradReducer
    torch.manual_seed(3)
    enc = ViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256)
    mae = MAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, decoder_dim_head=64).to(dev).train()
    g = torch.Generator().manual_seed(5)
    img = torch.randn(4, 3, 64, 64, generator=g).to(dev)
    idx = torch.rand(4, 16, generator=g).argsort(dim=-1).to(dev)
    loss0 = mae(img, rand_indices=idx)
    loss0.backward()
    ref = {k: p.grad.detach().clone() for k, p in mae.named_parameters() if p.grad is not None}
    for p in mae.parameters():
        p.grad = None
    red = GradReducer(mae, 1)
    assert mae.encoder.transformer._meta.sink is red and mae.decoder._meta.sink is red
    a0 = mae.encoder.transformer.layers[0][0]              # to_q / to_kv slots back to back: one dWqkv GEMM writes both
    assert red.slot(a0.to_kv.weight)[0] == sum(red.slot(a0.to_q.weight)) and red.target_block([a0.to_q.weight, a0.to_kv.weight]) is not None
    for step in range(2):                                   # the second pass must overwrite, not accumulate
        red.begin_step()
        loss = mae(img, rand_indices=idx)
        loss.backward()
        red.finish_step()
        assert abs(loss.item() - loss0.item()) < 1e-6 * max(1.0, abs(loss0.item()))
        for k, p in mae.named_parameters():
            if k not in ref:
                continue
            assert p.grad is not None and p.grad.data_ptr() == red._views[id(p)].data_ptr(), k     # still the flat-buffer view
            rel = ((p.grad - ref[k]).norm() / ref[k].norm().clamp_min(1e-30)).item()
            assert rel < 2e-6, (step, k, rel)
        missing = {id(p) for p in red.params_without_grad()}
        got_grad = {id(p) for k, p in mae.named_parameters() if k in ref}
        assert not (missing & got_grad)
    # evaluation with the sink attached: nothing is written, nothing breaks
    with torch.no_grad():
        mae.eval()
        assert torch.isfinite(mae(img, rand_indices=idx))


def test_mae_trainer_steps_with_fused_projection_as_a_view_of_the_flat_parameters(dev):
    """train.Trainer on the MAE wrapper: FusedAdamW keeps the parameters in a flat buffer with the gradient slots' layout,
    so [to_q.weight; to_kv.weight] is a zero-copy view of it (no per-step row-block copies) and its bf16 images follow the
    raw-pointer AdamW updates.  Three steps must give the losses of the same model trained WITHOUT the reducer sink and
    without FusedAdamW's layout: plain autograd + torch.optim.AdamW on a CPU-initialised twin, same inputs and mask."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from noise_robust_vit_amd.mae import MAE
    from noise_robust_vit_amd.train import TrainConfig, Trainer

    def make():
        torch.manual_seed(11)
        enc = ViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256)
        return MAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, decoder_dim_head=64).to(dev).train()

    g = torch.Generator().manual_seed(6)
    img = torch.randn(4, 3, 64, 64, generator=g).to(dev)
    idx = torch.rand(4, 16, generator=g).argsort(dim=-1).to(dev)
    a, b = make(), make()
    tr = Trainer(a, TrainConfig(lr=1e-2, weight_decay=0.05, grad_max_norm=0.0), None, compute_loss=lambda m, x, y: m(x, rand_indices=idx))
    att = a.encoder.transformer.layers[0][0]
    fused = att._refresh_fused()
    assert fused.data_ptr() == att.to_q.weight.data_ptr() and fused._base is None          # the view, not a copy
    assert torch.equal(fused[att.to_q.weight.shape[0]:], att.to_kv.weight.detach())
    opt = torch.optim.AdamW(b.parameters(), lr=1e-2, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999))
    for step in range(3):
        la = tr.step(img, None).item()
        opt.zero_grad(set_to_none=True)
        lb = b(img, rand_indices=idx)
        lb.backward()
        opt.step()
        from noise_robust_vit_amd.encoder import WEIGHTS
        WEIGHTS.clear()                                     # torch's fused AdamW may not bump version counters (train.py)
        assert abs(la - lb.item()) < 2e-3 * abs(lb.item()), (step, la, lb.item())
    assert att._refresh_fused().data_ptr() == att.to_q.weight.data_ptr()
    for (ka, pa), (kb, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pb.grad is None:
            continue
        rel = ((pa.detach() - pb.detach()).norm() / pb.detach().norm().clamp_min(1e-30)).item()
        assert rel < 2e-2, (ka, rel)          # three AdamW steps at lr 1e-2: sign-like updates amplify bf16-level gradient noise


def test_checkpoint_loaded_into_a_trained_lucid_model_reaches_the_fused_projection(dev):
    """Under Trainer + FusedAdamW the fused [to_q; to_kv] projection is a zero-copy view of the flat parameter buffer.  A
    checkpoint loaded AFTER a training step writes the parameters in place (version bump of the Parameters, not of the view):
    the next forward must use the loaded Q / K / V weights -- it must equal a fresh model loaded with the same checkpoint."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from noise_robust_vit_amd.encoder import WEIGHTS
    from noise_robust_vit_amd.train import TrainConfig, Trainer

    def make(seed):
        torch.manual_seed(seed)
        return ViT(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256).to(dev)

    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 64, 64, generator=g).to(dev)
    y = torch.randint(0, 10, (4,), generator=g).to(dev)
    other = {k: v.detach().clone() for k, v in make(21).state_dict().items()}
    m = make(20).train()
    tr = Trainer(m, TrainConfig(lr=1e-2, weight_decay=0.05, grad_max_norm=1.0))
    tr.step(x, y)                                        # stages the bf16 images of the fused view
    att = m.transformer.layers[0][0]
    assert att._refresh_fused().data_ptr() == att.to_q.weight.data_ptr()        # view mode
    m.load_state_dict(other)
    m.eval()
    with torch.no_grad():
        got = m(x)
    WEIGHTS.clear()
    fresh = make(22).eval()
    fresh.load_state_dict(other)
    with torch.no_grad():
        want = fresh(x)
    assert torch.equal(got, want), float((got - want).abs().max())


def test_lucid_fused_qkv_images_are_cached_and_follow_updates(dev):
    """lucid_vit.Attention feeds the fused QKV GEMM from a persistent [to_q; to_kv] buffer: no cast_transpose per forward
    once staged, and an in-place parameter update (version bump) or a raw-pointer update (FusedAdamW -> refresh_all) is
    picked up."""
    from noise_robust_vit_amd import encoder, kernels
    from noise_robust_vit_amd.lucid_vit import Attention
    torch.manual_seed(0)
    att = Attention(128, heads=2, dim_head=64).to(dev)
    x = torch.randn(2, 9, 128, device=dev)
    calls = []
    orig = kernels.cast_transpose
    kernels.cast_transpose = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            y0 = att(x)
            n0 = len(calls)
            y1 = att(x)
            assert len(calls) == n0 and torch.equal(y0, y1)          # second forward: every bf16 image comes from the cache
            att.to_q.weight.mul_(2.0)                                 # version bump -> fused buffer and image rebuilt
            y2 = att(x)
            assert len(calls) > n0 and not torch.equal(y2, y0)
            att.to_kv.weight.data.view(-1)[:] = att.to_kv.weight.data.view(-1) * 0.5     # raw update: no version bump of the Parameter
            encoder.WEIGHTS.refresh_all()
            y3 = att(x)
            assert not torch.equal(y3, y2)
    finally:
        kernels.cast_transpose = orig
    # gradients still reach the two source parameters
    att.zero_grad()
    att(x).sum().backward()
    assert att.to_q.weight.grad is not None and att.to_kv.weight.grad is not None
    assert att.to_q.weight.grad.shape == att.to_q.weight.shape


def test_noisy_input_training(dev):
    """`noise_std > 0` (the repo's namesake noisy-input training, examples/nowak.py:152-159: x + N(0, sigma^2) in front of
    the model): the perturbation reaches the model (first-step loss differs from the clean run with the same weights) and
    the HIP path still fits the batch through it."""
    from noise_robust_vit_amd import SimpleViT
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(32, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, 10, (32,), generator=g).to(dev)
    first = {}
    for std in (0.0, 0.5):
        torch.manual_seed(0)
        vit = SimpleViT(image_size=32, patch_size=8, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256).to(dev).train()
        tr = Trainer(vit, TrainConfig(lr=2e-3, weight_decay=0.05, grad_max_norm=5.0, label_smoothing=0.0, noise_std=std))
        torch.manual_seed(123)                                        # the noise stream
        first[std] = tr.step(x, y).item()
        if std > 0:
            for _ in range(80):
                last = tr.step(x, y).item()
    assert abs(first[0.0] - first[0.5]) > 1e-4, first
    assert last < 0.5 * first[0.5], (first, last)
