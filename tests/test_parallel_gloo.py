"""CPU, 2 processes, gloo: the data-parallel gradient reducer (noise_robust_vit_amd/parallel.py).

Checks the N > 1 protocol the GPU path uses with RCCL: flat gradient buffer, backward-ordered buckets, async
all-reduce launched per bucket as soon as its parameters are final, averaging, and that the reduced gradients equal
the single-process full-batch gradients (DDP semantics: per-rank batch = global / world, examples/CIFAR100.py:22).
A stand-in torch model is used (the HIP kernels need a GPU); the sink interface (`target` / `layer_done`) that the
encoder backward uses is exercised directly by a fake layer.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class SinkLayer(torch.autograd.Function):
    """y = x W^T whose backward writes dW straight into the reducer's buffer, like encoder.EncoderStackFn does."""

    @staticmethod
    def forward(ctx, x, w, sink):
        ctx.save_for_backward(x, w)
        ctx.sink = sink
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        out, beta = ctx.sink.target(w)
        g = dy.t() @ x
        out.copy_(g) if beta == 0.0 else out.add_(g)
        ctx.sink.layer_done(0, [w])
        return dy @ w, None, None


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.first = torch.nn.Linear(16, 32)
        self.w = torch.nn.Parameter(torch.randn(24, 32) * 0.1)       # sink-managed
        self.head = torch.nn.Linear(24, 5)
        self.sink = None

    def attach_grad_sink(self, sink):
        self.sink = sink

    def forward(self, x):
        h = torch.tanh(self.first(x))
        h = SinkLayer.apply(h, self.w, self.sink) if self.sink is not None else h @ self.w.t()
        return self.head(torch.tanh(h))


def _worker(rank, world, port, bucket_mib, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noise_robust_vit_amd.parallel import GradReducer
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 16, generator=g); Y = torch.randint(0, 5, (8,), generator=g)
    net = Net()
    red = GradReducer(net, world, bucket_mib=bucket_mib)
    tr = Trainer(net, TrainConfig(lr=1e-2, grad_max_norm=5.0), red)
    per = 8 // world
    xs, ys = X[rank * per:(rank + 1) * per], Y[rank * per:(rank + 1) * per]
    losses = []
    for _ in range(3):
        losses.append(tr.step(xs, ys).item())
    out = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # gradients of a fresh step, before the optimizer touches them
    tr.forward_backward(xs, ys)
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    if rank == 0:
        # plain numpy: torch tensors would travel as shared-memory handles that die with this process
        q.put(({k: v.numpy() for k, v in out.items()}, {k: v.numpy() for k, v in grads.items()}, red.bucket_bytes()))
    dist.barrier()
    dist.destroy_process_group()


def _reference():
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 16, generator=g); Y = torch.randint(0, 5, (8,), generator=g)
    net = Net()
    tr = Trainer(net, TrainConfig(lr=1e-2, grad_max_norm=5.0), None)
    for _ in range(3):
        tr.step(X, Y)
    out = {k: v.detach().clone() for k, v in net.state_dict().items()}
    tr.forward_backward(X, Y)
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    return out, grads


@pytest.mark.parametrize("bucket_mib", [64.0, 0.001])       # one bucket / several buckets
def test_two_rank_gloo_matches_single_process(bucket_mib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mib, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, grads, buckets = q.get(timeout=120)
    out = {k: torch.from_numpy(v) for k, v in out.items()}
    grads = {k: torch.from_numpy(v) for k, v in grads.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_out, ref_grads = _reference()
    for k in ref_grads:
        assert torch.allclose(grads[k], ref_grads[k], rtol=1e-5, atol=1e-6), k      # mean over ranks == full-batch grad
    for k in ref_out:
        assert torch.allclose(out[k], ref_out[k], rtol=1e-4, atol=1e-6), k          # 3 AdamW steps stay in lock-step
    if bucket_mib < 1:
        assert len(buckets) > 1
    assert sum(buckets) >= 4 * sum(p.numel() for p in Net().parameters())


def _worker_sync_eval(rank, world, port, q):
    """Per-rank seeds: replicas must still start from rank 0's weights (broadcast at construction, DDP semantics), and the
    evaluation accuracy is summed onto rank 0 with one scalar reduce (examples/CIFAR100.py:148-163)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noise_robust_vit_amd.parallel import GradReducer
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    net = Net()
    torch.manual_seed(1000 + rank)                      # replicas differ ...
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn_like(p) * 0.5)
    red = GradReducer(net, world)                       # ... until the reducer broadcasts rank 0's parameters
    tr = Trainer(net, TrainConfig(lr=1e-2, grad_max_norm=5.0), red)
    g = torch.Generator().manual_seed(5)
    X = torch.randn(8, 16, generator=g); Y = torch.randint(0, 5, (8,), generator=g)
    per = 8 // world
    xs, ys = X[rank * per:(rank + 1) * per], Y[rank * per:(rank + 1) * per]
    for _ in range(2):
        tr.step(xs, ys)
    w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    own = net(xs).argmax(1).eq(ys).float().mean().item()
    accs = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(accs, torch.tensor([own]))
    acc = tr.eval_step(xs, ys)
    mean_acc = tr.evaluate([(xs, ys)])
    if rank == 0:
        q.put((max((gathered[0] - t).abs().max().item() for t in gathered), acc.item(), sum(a.item() for a in accs), mean_acc))
    dist.barrier()
    dist.destroy_process_group()


def test_param_broadcast_and_eval_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sync_eval, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    spread, acc0, acc_sum, mean_acc = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert spread == 0.0, spread                        # identical replicas after two steps despite per-rank seeds
    assert abs(acc0 - acc_sum) < 1e-6                   # rank 0 holds the SUM of the per-rank accuracies (dist.reduce, dst 0)
    assert abs(mean_acc - acc_sum / 2) < 1e-6


def _worker_forced(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from noise_robust_vit_amd.parallel import GradReducer
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 16, generator=g); Y = torch.randint(0, 5, (8,), generator=g)
    net = Net()
    red = GradReducer(net, 1, bucket_mib=0.001, force_collectives=True)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), orig(*a, **k))[1]
    tr = Trainer(net, TrainConfig(lr=1e-2, grad_max_norm=5.0), red)
    for _ in range(3):
        tr.step(X, Y)
    out = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
    q.put((out, len(calls), len(red.buckets)))
    dist.destroy_process_group()


def test_forced_collectives_world_one_equal_plain_run():
    """world_size 1 with force_collectives: every bucket issues its all_reduce (the code path of the multi-GPU step) and
    the training trajectory equals the run without a process group bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_forced, args=(_free_port(), q))
    p.start()
    out, ncalls, nbuckets = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert nbuckets > 1 and ncalls == 3 * nbuckets
    ref_out, _ = _reference()
    for k in ref_out:
        assert torch.equal(torch.from_numpy(out[k]), ref_out[k]), k


def _worker_bf16(rank, world, port, q):
    """grad_dtype="bf16": each rank's bucket is rounded to bf16, reduced, and converted back; tail_mib caps the last bucket."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noise_robust_vit_amd.parallel import GradReducer
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 16, generator=g); Y = torch.randint(0, 5, (8,), generator=g)
    net = Net()
    _ = net(X[:2])                                       # a forward BEFORE the reducer exists (warm-up / sanity pass)
    red = GradReducer(net, world, bucket_mib=64.0, grad_dtype="bf16", tail_mib=0.0005)
    tr = Trainer(net, TrainConfig(lr=1e-2, grad_max_norm=5.0), red)
    per = 8 // world
    xs, ys = X[rank * per:(rank + 1) * per], Y[rank * per:(rank + 1) * per]
    tr.forward_backward(xs, ys)
    grads = {k: p.grad.detach().clone().numpy() for k, p in net.named_parameters()}
    if rank == 0:
        q.put((grads, red.bucket_bytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_exchange_cost_and_tail_bucket():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    grads, buckets = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, ref_grads = _reference_grads_only()
    worst = 0.0
    for k, ref in ref_grads.items():
        got = torch.from_numpy(grads[k])
        # the stated numerical cost: every addend rounded to 8 significant bits before the mean, the mean rounded once more
        assert (got - ref).abs().max() <= 2.0 ** -7 * ref.abs().max() + 1e-7, k
        worst = max(worst, ((got - ref).norm() / ref.norm()).item())
    assert worst < 6e-3, worst                           # measured: ~2e-3 relative L2 (bf16 epsilon 3.9e-3)
    # tail cap 0.0005 MiB = 131 floats: the last bucket is cut down to the last parameter of the backward order alone
    # (Net.w, 24 x 32 floats -- larger than the cap by itself, and a bucket never splits a parameter)
    assert len(buckets) == 2 and buckets[-1] == 4 * 24 * 32


def _reference_grads_only():
    from noise_robust_vit_amd.train import TrainConfig, Trainer
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 16, generator=g); Y = torch.randint(0, 5, (8,), generator=g)
    net = Net()
    tr = Trainer(net, TrainConfig(lr=1e-2, grad_max_norm=5.0), None)
    tr.forward_backward(X, Y)
    return None, {k: p.grad.detach().clone() for k, p in net.named_parameters()}
