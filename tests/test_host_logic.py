"""CPU: host-side logic of the drop-in modules -- constructor signatures, state_dict contract, initialisers,
checkpoint interop, LR schedule -- and the rule that the product never touches the oracle or a CPU fallback."""
import ast
import math
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_simplevit_state_dict_and_seeded_init_equal_the_reference(golden_dir):
    """Same module construction order as simple_vit.py => same RNG consumption => identical weights under a seed."""
    from noise_robust_vit_amd import SimpleViT
    g = np.load(f"{golden_dir}/simplevit_cfg1_weights.npz")
    torch.manual_seed(0)
    m = SimpleViT(image_size=32, patch_size=16, num_classes=100, dim=192, depth=2, heads=3, mlp_dim=768)
    sd = m.state_dict()
    assert sorted(sd) == sorted(g.keys())
    for k in g.keys():
        assert np.array_equal(sd[k].numpy(), g[k]), k
    m2 = SimpleViT(image_size=32, patch_size=16, num_classes=100, dim=192, depth=2, heads=3, mlp_dim=768, robust=True)
    assert sorted(m2.state_dict()) == sorted(sd)                        # SinkhornAttention has no parameters
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in g.items()})


def test_vision_transformer_keys_shapes_and_init():
    from noise_robust_vit_amd.vit import VisionTransformer, vit_b_16
    from oracle import vit_oracle as V
    cfg = dict(image_size=32, patch_size=16, num_layers=2, num_heads=3, hidden_dim=192, mlp_dim=768, num_classes=10)
    m = VisionTransformer(**cfg)
    ref = V.vit_init_state_dict(seed=0, **cfg)
    sd = m.state_dict()
    assert sorted(sd) == sorted(ref)                                     # SURVEY.md §8b key contract
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert sd["heads.head.weight"].abs().max() == 0 and sd["class_token"].abs().max() == 0     # vit.py:247,304-306
    blk = "encoder.layers.encoder_layer_0."
    assert sd[blk + "self_attention.in_proj_bias"].abs().max() == 0                              # utils.py:727
    assert sd[blk + "mlp.0.bias"].abs().max() < 1e-4                                             # vit.py:53
    assert abs(sd["encoder.pos_embedding"].std().item() - 0.02) < 0.003
    b = vit_b_16()
    assert sum(p.numel() for p in b.parameters()) == 86567656           # torchvision vit_b_16 parameter count
    m.load_state_dict(ref)


def test_legacy_mlp_keys_load():
    from noise_robust_vit_amd.vit import VisionTransformer
    m = VisionTransformer(image_size=32, patch_size=16, num_layers=1, num_heads=1, hidden_dim=64, mlp_dim=128, num_classes=3)
    legacy = {k.replace(".mlp.0.", ".mlp.linear_1.").replace(".mlp.3.", ".mlp.linear_2."): v.clone() + 1.0
              for k, v in m.state_dict().items()}
    m.load_state_dict(legacy)                                            # vit.py:55-84
    assert torch.equal(m.encoder.layers.encoder_layer_0.mlp[3].bias.data,
                       legacy["encoder.layers.encoder_layer_0.mlp.linear_2.bias"])


def test_cpu_forward_fails_loudly_no_fallback():
    from noise_robust_vit_amd import Attention, SimpleViT
    from noise_robust_vit_amd._lib import NrvError
    m = SimpleViT(image_size=32, patch_size=16, num_classes=10, dim=64, depth=1, heads=1, mlp_dim=128)
    with pytest.raises(NrvError, match="no CPU fallback"):
        m(torch.randn(2, 3, 32, 32))
    with pytest.raises(NrvError):
        Attention(64, heads=1)(torch.randn(2, 4, 64))


def test_sincos_table_matches_oracle():
    from noise_robust_vit_amd.simple_vit import sincos_table_2d
    from oracle.simple_vit_oracle import posemb_sincos_2d
    for h, w, d in ((2, 2, 192), (14, 14, 768), (3, 5, 64)):
        assert (sincos_table_2d(h, w, d) - posemb_sincos_2d(h, w, d)).abs().max() < 1e-6


def test_warmup_cosine_lr_matches_torch_schedulers():
    """Closed form vs SequentialLR([LinearLR(1e-3,1,T1), CosineAnnealingLR(T2, eta_min=0.05 lr)]) (CIFAR100.py:99-113)."""
    from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR
    from noise_robust_vit_amd.train import warmup_cosine_lr
    base, T1, T2 = 5e-4, 7, 31
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=base)
    sched = SequentialLR(opt, [LinearLR(opt, 1e-3, 1, total_iters=T1), CosineAnnealingLR(opt, T_max=T2, eta_min=base * 0.05)],
                         milestones=[T1])
    for step in range(T1 + T2):
        assert math.isclose(opt.param_groups[0]["lr"], warmup_cosine_lr(step, base, T1, T2), rel_tol=1e-6, abs_tol=1e-12), step
        opt.step(); sched.step()


def _imports(path):
    tree = ast.parse(open(path).read())
    out = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            out |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
            out.add(node.module.split(".")[0])
    return out


def _non_doc_strings_with(path, needle):
    tree = ast.parse(open(path).read())
    docs = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef)) and node.body and \
                isinstance(node.body[0], ast.Expr) and isinstance(node.body[0].value, ast.Constant):
            docs.add(id(node.body[0].value))
    return [n.value for n in ast.walk(tree)
            if isinstance(n, ast.Constant) and isinstance(n.value, str) and needle in n.value and id(n) not in docs]


def test_product_never_imports_the_oracle_or_reads_the_reference():
    pkg = os.path.join(ROOT, "noise_robust_vit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                path = os.path.join(dirpath, f)
                assert "oracle" not in _imports(path), path
                assert not _non_doc_strings_with(path, "/root/reference"), path     # citations in docstrings are fine
    # bench.py may use the oracle only inside cpu_baseline(); smoke() only as a checker
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name != "cpu_baseline":
            for sub in ast.walk(node):
                if isinstance(sub, ast.ImportFrom) and sub.module and sub.module.startswith("oracle"):
                    raise AssertionError(f"bench.py:{node.name} imports the oracle")


def test_robust_flag_is_plumbed_not_substituted():
    """robust=True must never silently run softmax attention."""
    from noise_robust_vit_amd import SimpleViT
    from noise_robust_vit_amd.simple_vit import SinkhornAttention
    m = SimpleViT(image_size=32, patch_size=16, num_classes=10, dim=64, depth=1, heads=1, mlp_dim=128, robust=True)
    assert isinstance(m.transformer.layers[0][0].attend, SinkhornAttention)
    assert m.transformer._meta.robust and m.transformer.layers[0][0]._meta.robust


def test_interpolate_embeddings_matches_direct_resample():
    """vit.py:522-603: class-token position untouched, the 2x2 patch grid of a 32-px checkpoint resampled to 4x4."""
    from collections import OrderedDict
    from noise_robust_vit_amd.vit import VisionTransformer, interpolate_embeddings
    src = VisionTransformer(image_size=32, patch_size=16, num_layers=1, num_heads=1, hidden_dim=64, mlp_dim=128, num_classes=3)
    sd = OrderedDict((k, v.clone()) for k, v in src.state_dict().items())
    pos = sd["encoder.pos_embedding"].clone()
    out = interpolate_embeddings(64, 16, sd, reset_heads=True)
    assert not any(k.startswith("heads") for k in out) and "encoder.ln.weight" in out
    new = out["encoder.pos_embedding"]
    assert new.shape == (1, 17, 64) and torch.equal(new[:, 0], pos[:, 0])
    grid = pos[0, 1:].t().reshape(1, 64, 2, 2)
    ref = torch.nn.functional.interpolate(grid, size=4, mode="bicubic", align_corners=True).reshape(64, 16).t()
    assert torch.allclose(new[0, 1:], ref, atol=1e-6)
    # corners are preserved with align_corners=True
    assert torch.allclose(new[0, 1], pos[0, 1], atol=1e-6) and torch.allclose(new[0, 16], pos[0, 4], atol=1e-6)
    dst = VisionTransformer(image_size=64, patch_size=16, num_layers=1, num_heads=1, hidden_dim=64, mlp_dim=128, num_classes=3)
    missing = dst.load_state_dict(out, strict=False)
    assert set(missing.missing_keys) == {"heads.head.weight", "heads.head.bias"} and not missing.unexpected_keys
    # same size: state returned untouched
    same = interpolate_embeddings(32, 16, OrderedDict(src.state_dict()))
    assert same["encoder.pos_embedding"].shape == (1, 5, 64)


def test_cutmix_box_and_trainer_loss_mix():
    """CIFAR100.py:119-137: box area ~ (1 - lam), loss = lam CE(y) + (1 - lam) CE(y[perm]) with lam = exact pixel ratio."""
    import numpy as np
    from noise_robust_vit_amd.train import TrainConfig, Trainer, cutmix_box
    rng = np.random.default_rng(0)
    for lam in (0.1, 0.5, 0.9):
        for _ in range(20):
            a1, b1, a2, b2 = cutmix_box(224, 224, lam, rng)
            assert 0 <= a1 <= a2 <= 224 and 0 <= b1 <= b2 <= 224
            assert (a2 - a1) * (b2 - b1) <= (1 - lam) * 224 * 224 + 1
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(3 * 8 * 8, 5))
    x = torch.randn(6, 3, 8, 8); y = torch.randint(0, 5, (6,))
    tr = Trainer(net, TrainConfig(cutmix_prob=1.0, seed=3, grad_max_norm=0.0))
    torch.manual_seed(11)
    loss = tr.forward_backward(x, y)
    # replay the same random decisions
    rng = np.random.default_rng(3); assert rng.random() < 1.0
    torch.manual_seed(11); perm = torch.randperm(6)
    lam = float(rng.beta(1.0, 1.0)); a1, b1, a2, b2 = cutmix_box(8, 8, lam, rng)
    xm = x.clone(); xm[:, :, a1:a2, b1:b2] = x[perm, :, a1:a2, b1:b2]
    lam = 1.0 - (a2 - a1) * (b2 - b1) / 64.0
    out = net(xm)
    ce = lambda t: torch.nn.functional.cross_entropy(out, t, label_smoothing=0.1)
    assert abs(loss.item() - (ce(y) * lam + ce(y[perm]) * (1 - lam)).item()) < 1e-6


def test_trainer_on_cpu_uses_torch_adamw_and_fused_optimizer_refuses_cpu():
    """The fused optimizer is a HIP path: on CPU tensors the Trainer falls back to torch.optim.AdamW (whose arithmetic the
    HIP kernels are pinned against on the GPU), and FusedAdamW itself refuses a CPU gradient buffer."""
    from noise_robust_vit_amd._lib import NrvError
    from noise_robust_vit_amd.optim import FusedAdamW
    from noise_robust_vit_amd.parallel import GradReducer
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    net = torch.nn.Linear(4, 3)
    tr = Trainer(net, TrainConfig(grad_max_norm=1.0))
    assert isinstance(tr.opt, torch.optim.AdamW) and not tr.fused
    w0 = net.weight.detach().clone()
    tr.step(torch.randn(5, 4), torch.randint(0, 3, (5,)))
    assert not torch.equal(w0, net.weight.detach())
    red = GradReducer(torch.nn.Linear(4, 3), 1)
    assert red.parameters() and red.slot(red.parameters()[0])[1] == 12
    with pytest.raises(NrvError):
        FusedAdamW(red, lr=1e-3)


def test_grad_reducer_places_grouped_parameters_back_to_back():
    """`grad_groups()` (lucid_vit: to_q / to_kv, whose gradients are the row blocks of one fused-projection GEMM): the
    reducer lays the group's slots out consecutively in the group's order, `target_block` hands out ONE view over them,
    every other parameter keeps backward order, and an MAE / lucid model exposes one group per attention layer."""
    from noise_robust_vit_amd.lucid_vit import ViT
    from noise_robust_vit_amd.mae import MAE
    from noise_robust_vit_amd.parallel import GradReducer
    enc = ViT(image_size=32, patch_size=16, num_classes=4, dim=64, depth=2, heads=1, mlp_dim=128)
    mae = MAE(encoder=enc, decoder_dim=64, decoder_depth=1, decoder_heads=1, decoder_dim_head=64)
    groups = mae.grad_groups()
    assert len(groups) == 3 and all(len(g) == 2 for g in groups)
    red = GradReducer(mae, 1, attach=False)
    used = 0
    for q, kv in groups:
        (oq, nq), (okv, nkv) = red.slot(q), red.slot(kv)
        assert okv == oq + nq                                  # to_kv directly behind to_q
        blk = red.target_block([q, kv])
        assert blk is not None and blk[0].shape == (q.shape[0] + kv.shape[0], q.shape[1]) and blk[1] == 0.0
        assert blk[0].data_ptr() == q.grad.data_ptr()
        used += 1
    assert red.target_block([groups[0][1], groups[0][0]]) is None          # wrong order: not one block
    # slots tile the flat buffer without overlap
    spans = sorted(red.slot(p) for p in red.parameters())
    for (o0, n0), (o1, _) in zip(spans, spans[1:]):
        assert o0 + (n0 + 7) // 8 * 8 == o1                     # slots of parallel.SLOT_ALIGN = 8 elements
    assert spans[-1][0] + (spans[-1][1] + 7) // 8 * 8 == red.flat.numel()


def test_begin_step_releases_a_cu_reservation_left_by_an_aborted_backward(monkeypatch):
    """the CU reservation is process-wide; a backward that raises after the first bucket never reaches finish_step,
    so the next begin_step must put the GEMM planning back to all CUs."""
    from noise_robust_vit_amd import kernels as K
    from noise_robust_vit_amd.parallel import GradReducer
    calls = []
    monkeypatch.setattr(K, "set_reserved_cus", lambda n: calls.append(n) or 0)
    red = GradReducer(torch.nn.Linear(4, 3), 1)
    red._reserved = True                      # what _launch leaves behind when the step never finishes
    red.begin_step()
    assert calls == [0] and red._reserved is False
    red.begin_step()
    assert calls == [0]                       # nothing to release: no second call


def test_begin_step_clears_autograd_slots_as_contiguous_ranges():
    """The autograd-managed gradient slots are zeroed as maximal runs of the flat buffer (not one fill per
    parameter), and the runs follow the set of kernel-written parameters."""
    from noise_robust_vit_amd.parallel import GradReducer
    net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
    red = GradReducer(net, 1)
    assert len(red._autograd_ranges()) == 1 and red._autograd_ranges()[0].numel() == red.flat.numel()
    red.flat.fill_(3.0)
    w_mid = net[1].weight
    red.target(w_mid)                         # a kernel writes this one: its slot is not cleared by begin_step
    red.begin_step()
    o, n = red.slot(w_mid)
    assert len(red._autograd_ranges()) == 2
    assert torch.all(red.flat[o:o + n] == 3.0)
    mask = torch.ones_like(red.flat, dtype=torch.bool)
    mask[o:o + n] = False
    assert torch.all(red.flat[mask] == 0.0)
    assert all(p.grad is red._views[id(p)] for p in net.parameters())


def test_capture_recorder_names_the_replay_unsafe_library_ops():
    """Trainer.capture refuses steps by the NAME of a replay-unsafe PyTorch op (round 3: nn.Embedding's sort-based backward faulted
    inside rocprim's partition kernel in the first replay), not by `compute_loss is not None`: the recorder sees forward and
    backward ops of one eager step."""
    from noise_robust_vit_amd.train import _record_unsafe_ops
    emb = torch.nn.Embedding(10, 4)

    def bad():
        w = torch.zeros(5, 4, requires_grad=True)
        (emb(torch.randint(0, 10, (6,))).sum() + w[torch.tensor([1, 1, 3])].sum()).backward()

    def fine():
        x = torch.randn(4, 4, requires_grad=True)
        idx = torch.tensor([[0, 2], [1, 3]])
        (x * 2).gather(1, idx).sum().backward()           # gather / scatter_add: no selection primitive

    assert _record_unsafe_ops(bad) == {"aten.embedding_dense_backward", "aten.index_put(accumulate=True)"}
    assert _record_unsafe_ops(fine) == set()
