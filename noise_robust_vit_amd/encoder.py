"""Encoder-block hot path: forward/backward schedules over the HIP kernels, exposed as autograd Functions.

One transformer block = two "halves", both pre-LayerNorm with a residual add:

    attention half :  x -> LN -> QKV GEMM -> fused attention -> out-proj GEMM (+bias) (+x)
    MLP half       :  x -> LN -> fc1 GEMM +bias +GELU -> fc2 GEMM +bias (+x)

which is `Transformer.forward` of the reference (simple_vit.py:93-97: x = attn(x) + x; x = ff(x) + x) and
`EncoderBlock.forward` (vit.py:118-130).  The residual stream is fp32 in HBM; every GEMM operand is bf16;
LayerNorm statistics, softmax and all accumulation are fp32 (DESIGN.md "Numerics").

Backward is hand-scheduled (no autograd graph inside the stack): per half,
    dW2 = dY^T H (TN GEMM), dH = dY W2 (NT GEMM on the staged W2^T) with the GELU' epilogue, ...
and the LayerNorm backward kernel adds the residual-stream gradient and emits both the fp32 stream
gradient and the bf16 copy the next GEMMs consume.

Nothing here computes with torch ops: torch supplies memory, streams and the autograd boundary.
"""
from __future__ import annotations

import functools
import weakref
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch

from . import kernels as K
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_Q8, EPI_BIAS_RESIDUAL, EPI_DGELU, EPI_DGELU_Q8, EPI_NONE, NrvError)

Tensor = torch.Tensor

PARAMS_PER_LAYER = 12   # ln1_w ln1_b wqkv bqkv wo bo ln2_w ln2_b w1 b1 w2 b2


# ----------------------------------------------------------------------------------------------
# bf16 weight staging: W (fp32 master, nn.Parameter) -> bf16 W and bf16 W^T, refreshed when the
# parameter's version counter changes (i.e. once per optimizer step).
# ----------------------------------------------------------------------------------------------
class WeightCache:
    """(id(owner), data_ptr, shape) -> (weakref(owner), version, bf16 W, bf16 W^T); owner = the tensor, or its base when
    it is a view of a persistent buffer (lucid_vit.Attention's fused [to_q; to_kv] projection).  Dead entries are purged.

    `derived` sources: fp32 buffers that are COPIES of parameters (the fused projection above) register a refresher; it
    runs before `refresh_all()` re-stages the images and whenever `epoch` moved, so a copy never outlives an update of
    its parameters through raw pointers (which leaves version counters untouched)."""

    def __init__(self) -> None:
        self._d = {}
        self._jobs = None
        self._derived = []           # weak references to bound methods
        self.epoch = 0

    @staticmethod
    def _owner(w: Tensor) -> Tensor:
        return w._base if w._base is not None else w

    def get(self, w: Tensor, need_t: bool):
        owner = self._owner(w)
        key = (id(owner), w.data_ptr(), tuple(w.shape))
        ent = self._d.get(key)
        ver = w._version
        if ent is not None and ent[0]() is owner and ent[1] == ver:
            return ent[2], ent[3]
        w2 = w.detach()
        if w2.dim() != 2:                       # Conv2d patch-embed weight [D, C, p, p] viewed as [D, C*p*p]
            w2 = w2.reshape(w2.shape[0], -1)
        wb, wt = K.cast_transpose(w2, need_t=True)
        if len(self._d) > 2048:
            self._d = {k: v for k, v in self._d.items() if v[0]() is not None}
        self._d[key] = (weakref.ref(owner), ver, wb, wt)
        self._jobs = None
        return wb, wt

    def add_derived(self, bound_method) -> None:
        self._derived.append(weakref.WeakMethod(bound_method))

    def _run_derived(self) -> None:
        alive = []
        for r in self._derived:
            fn = r()
            if fn is not None:
                fn()
                alive.append(r)
        self._derived = alive

    def clear(self) -> None:
        self._d.clear()
        self._jobs = None
        self.epoch += 1

    def refresh_all(self) -> None:
        """Re-stage every live entry IN PLACE with one batched launch (after an optimizer step that updated the fp32 masters
        through raw pointers: version counters and addresses are unchanged, so the entries stay valid afterwards)."""
        self.epoch += 1
        self._run_derived()                    # fp32 copies of parameters first: their images are re-staged below
        live = [(k, v) for k, v in self._d.items() if v[0]() is not None]
        if len(live) != len(self._d):
            self._d = dict(live)
            self._jobs = None
        key = tuple(k for k, _ in live)
        jobs = self._jobs
        if jobs is None or jobs[0] != key:
            srcs = []
            for k, (ref, ver, wb, wt) in live:
                w = ref().detach()
                if w.data_ptr() != k[1] or w.numel() != wb.numel() or not w.is_contiguous():
                    self.clear()                 # something else changed (or a partial view): fall back to lazy re-staging
                    return
                srcs.append((w.reshape(wb.shape), wb, wt))
            jobs = (key, K.build_cast_jobs(srcs), srcs)
            self._jobs = jobs
        K.run_cast_jobs(jobs[1])
        # the images are current again: entries whose source bumped its version meanwhile (derived copies) stay valid
        for k, v in list(self._d.items()):
            o = v[0]()
            if o is not None:
                self._d[k] = (v[0], o._version, v[2], v[3])


WEIGHTS = WeightCache()


@dataclass
class BlockMeta:
    heads: int
    dim_head: int
    eps: float
    robust: bool = False
    # optional gradient sink (data-parallel runtime): grad_out(param) -> (tensor, beta) or None
    sink: Optional[object] = None
    # residual / MLP dropout of the block stack (vit.py:100-101,112,125): probability for the NEXT forward (the modules set it
    # from their own p and training flag) and an optional source of keep masks, site -> uint8 tensor (tests inject the masks the
    # oracle uses; default: torch's device generator).  Sites of layer i: 3 i (attention branch), 3 i + 1 (GELU output), 3 i + 2 (MLP branch).
    dropout: float = 0.0
    mask_source: Optional[object] = None
    # dropout on the attention weights (vit.py:108): composed on the materialised [B,H,N,N] matrix (kernels.attn_dropout_fwd), site -(2 + i)
    attn_dropout: float = 0.0
    # additive score bias of ONE call (attention / key-padding masks of the stand-alone MultiheadAttention, utils.py:741-751): fp32,
    # broadcastable to [B, H, N, N], -inf = masked; composed path
    attn_bias: Optional[Tensor] = None


def _grad_target(meta: BlockMeta, p: Optional[Tensor]):
    """(out, beta) for a weight gradient: the sink's flat-buffer view if a sink is attached, else fresh."""
    if p is None:
        return None, 0.0
    if meta.sink is not None:
        return meta.sink.target(p)
    return None, 0.0


def sink_params(params) -> list:
    """The registered parameters behind a layer's tensors: a fused projection (`_nrv_parts`, lucid_vit.Attention) stands for
    the parameters whose rows it stacks."""
    out = []
    for q in params:
        if q is None:
            continue
        parts = getattr(q, "_nrv_parts", None)
        out += [pp for pp, _, _ in parts] if parts is not None else [q]
    return out


# Weight gradients on a second HIP stream.  dW = dy^T x only feeds the optimizer, so it can run beside the dX chain (the NT GEMM
# of the same layer reads the same dy) and take the CUs that chain leaves idle: with one 512-thread workgroup per CU a grid
# of 316 or 396 tiles (ViT-L, ViT-S, the MAE encoder) occupies 62 - 77 % of its CU-rounds.  Measured on the whole step
# (profiles/r03_wgrad_stream_ab.txt): ViT-S -2.5 %, ViT-L -2.1 %, MAE -2.6 %; ViT-B/16 at batch 256 (474 tiles, 93 %) +-0.1 %,
# where the chip is already power-limited -- so the stream is used only when the dX grid fills its rounds poorly ("auto").
WGRAD_STREAM = "auto"            # "auto" | True | False
WGRAD_MIN_IDLE = 0.15
_WGRAD_SIDE = {}
_WGRAD_KEEP: list = []


def _wgrad_side(device) -> "torch.cuda.Stream":
    s = _WGRAD_SIDE.get(device.index)
    if s is None:
        s = _WGRAD_SIDE[device.index] = torch.cuda.Stream(device=device)
    return s


@functools.lru_cache(maxsize=None)
def _dx_grid_idle(rows: int, cols: int, cus: int) -> float:
    """Share of CU-rounds the [rows x cols] dX GEMM leaves empty under the best of the NT kernel's tile heights (256-column tiles)."""
    best = 1.0
    for h in (256, 320, 192, 128):
        tiles = -(-rows // h) * -(-cols // 256)
        best = min(best, 1.0 - tiles / (-(-tiles // cus) * cus))
    return best


def _wgrad_on_side(dy16: Tensor, x16: Tensor) -> bool:
    if K._PROF is not None:                  # per-launch timing (bench.py's roofline leg): one stream, every launch timed alone
        return False
    if WGRAD_STREAM == "auto":
        # judged on the layer's narrow side (the residual width: every second GEMM of the dX chain has that many columns);
        # deciding per GEMM instead was measured 0.5 - 1 % slower on ViT-S / ViT-L than moving the whole layer's weight gradients
        cus = torch.cuda.get_device_properties(x16.device).multi_processor_count
        return _dx_grid_idle(x16.shape[0], min(x16.shape[1], dy16.shape[1]), cus) >= WGRAD_MIN_IDLE
    return bool(WGRAD_STREAM)


def wgrad_join() -> None:
    """The current stream waits for every weight gradient issued so far; their operands may be freed afterwards."""
    if _WGRAD_KEEP:
        dev = _WGRAD_KEEP[0][0].device
        torch.cuda.current_stream(dev).wait_stream(_wgrad_side(dev))
        _WGRAD_KEEP.clear()


# Weight gradients are QUEUED while a layer is differentiated and issued together when it is done (`wgrad_flush`): the four of an
# encoder block (dWqkv, dWo, dW1, dW2 and their bias gradients) are ONE stream-K launch (kernels.gemm_tn_grouped) that fills every
# CU whatever the matrices' tile counts and pays prologue, epilogue and partial-tile traffic once per layer instead of once per
# gradient (round 4; before: one split-K launch + one slab reduction per gradient).
_PENDING: list = []
WGRAD_GROUPED = False            # True: one grouped launch per layer (kernels.gemm_tn_grouped); False: every gradient is its own split-K launch, issued where it is queued (the round-3 schedule)


def _queue(meta: BlockMeta, q: dict) -> None:
    _PENDING.append(q)
    if not WGRAD_GROUPED:
        wgrad_flush(meta)


def _dw_db(meta: BlockMeta, dy16: Tensor, x16: Tensor, w: Tensor, b: Optional[Tensor]):
    """Queue the weight and bias gradient of y = x W^T + b (dW = dy^T x on the MFMA, db = colsum(dy) fused into the same kernel).
    Returns (dW, db) as the tensors the launch of `wgrad_flush` will fill: the sink's flat-buffer views, or fresh ones."""
    parts = getattr(w, "_nrv_parts", None)
    if parts is not None and meta.sink is not None:
        # w stacks the rows of several parameters ([to_q; to_kv]): straight into their slots of the sink -- one problem when the
        # slots are adjacent (GradReducer + grad_groups), else one per parameter on the matching column block of dy
        if b is not None:
            raise NrvError("a fused projection carries no bias")
        block = meta.sink.target_block([pp for pp, _, _ in parts]) if hasattr(meta.sink, "target_block") else None
        if block is not None:
            _queue(meta, dict(A=dy16, B=x16, out=block[0], beta=block[1]))
            return None, None
        for pp, r0, r1 in parts:
            tw, bw = meta.sink.target(pp)
            _queue(meta, dict(A=dy16[:, r0:r1], B=x16, out=tw, beta=bw))
        return None, None
    tw, bw = _grad_target(meta, w)
    tb, bb = _grad_target(meta, b) if b is not None else (None, 0.0)
    M, N = dy16.shape[1], x16.shape[1]
    out = tw if tw is not None else torch.empty(M, N, dtype=torch.float32, device=dy16.device)
    dbias = None if b is None else (tb if tb is not None else torch.empty(M, dtype=torch.float32, device=dy16.device))
    _queue(meta, dict(A=dy16, B=x16, out=out, beta=bw if tw is not None else 0.0, dbias=dbias, dbias_beta=bb if tb is not None else 0.0))
    return out, dbias


def wgrad_flush(meta: BlockMeta) -> None:
    """Issue the queued weight gradients (groups of up to 4 with the same token count).  With a gradient sink (results go to
    persistent buffers nobody reads before `wgrad_join`) the launch may go to the side stream, beside the next layer's dX chain."""
    if not _PENDING:
        return
    probs = list(_PENDING)
    _PENDING.clear()
    groups, cur = [], []
    for q in probs:
        if cur and (len(cur) == 4 or q["A"].shape[0] != cur[0]["A"].shape[0]):
            groups.append(cur)
            cur = []
        cur.append(q)
    groups.append(cur)
    dy16, x16 = probs[0]["A"], probs[0]["B"]
    if meta.sink is not None and _wgrad_on_side(dy16, x16):
        side = _wgrad_side(dy16.device)
        side.wait_stream(torch.cuda.current_stream(dy16.device))
        _WGRAD_KEEP.append(tuple(t for q in probs for t in (q["A"], q["B"])))     # alive until the join
        with torch.cuda.stream(side):
            for g in groups:
                _issue(g)
        return
    for g in groups:
        _issue(g)


def _issue(group: list) -> None:
    if WGRAD_GROUPED:
        K.gemm_tn_grouped(group)
        return
    for q in group:
        K.gemm_tn(q["A"], q["B"], out=q["out"], beta=q["beta"], dbias=q.get("dbias"), dbias_beta=q.get("dbias_beta", 0.0))


# ----------------------------------------------------------------------------------------------
# introspection: attention maps for recorder.Recorder (the fused kernels never materialise them)
# ----------------------------------------------------------------------------------------------
_RECORDING: Optional[list] = None


class record_attention:
    """Context manager: every attention layer run inside appends its fp32 [B, H, N, N] probabilities to `sink`
    (softmax, or the Sinkhorn-normalised matrix for robust=True), recomputed from q, k and the saved statistics."""

    def __init__(self, sink: list) -> None:
        self.sink = sink

    def __enter__(self):
        global _RECORDING
        self._prev, _RECORDING = _RECORDING, self.sink
        return self.sink

    def __exit__(self, *exc):
        global _RECORDING
        _RECORDING = self._prev


def _record(qkv: Tensor, aux, B: int, N: int, H: int, dh: int, scale: float, robust: bool) -> None:
    if robust:
        lse, scal = aux[0], aux[1]                        # scalings [B, H, 7, N]: a1 b1 a2 b2 a3 b3 a4
        p = K.attn_probs(qkv, lse, B, N, H, dh, scale)
        a, b = scal[:, :, 6], scal[:, :, 5]              # the vectors are cumulative: P = diag(a4) softmax(S) diag(b3)
        p = p * a[..., :, None] * b[..., None, :]
    else:
        p = K.attn_probs(qkv, aux, B, N, H, dh, scale)
    _RECORDING.append(p)


# ----------------------------------------------------------------------------------------------
# attention half
# ----------------------------------------------------------------------------------------------
def draw_keep(meta: BlockMeta, site: int, shape, device, p: Optional[float] = None) -> Tensor:
    """uint8 keep mask of one dropout site (1 = kept); p defaults to the stack's `dropout`."""
    if meta.mask_source is not None:
        m = meta.mask_source(site, tuple(shape))
        if m.dtype != torch.uint8 or tuple(m.shape) != tuple(shape):
            raise NrvError(f"mask_source({site}): expected a uint8 mask of shape {tuple(shape)}")
        return m.to(device).contiguous()
    return (torch.rand(shape, device=device) >= (meta.dropout if p is None else p)).to(torch.uint8)


def attn_half_fwd(x: Tensor, B: int, N: int, meta: BlockMeta, ln_w, ln_b, wqkv, bqkv, wo, bo, residual: bool, drop=None, adrop=None):
    """x fp32 [B*N, D] -> (y fp32 [B*N, D], saved).  `ln_w is None`: no LayerNorm in front of the projection (the bare
    `MultiheadAttention.forward` of the reference's forked module, utils.py:741-751)."""
    H, dh = meta.heads, meta.dim_head
    if ln_w is None:
        xn, mean, rstd = K.cast_bf16(x), None, None
    else:
        xn, mean, rstd = K.layernorm_fwd(x, ln_w, ln_b, meta.eps)
    wqkv_b, _ = WEIGHTS.get(wqkv, True)
    wo_b, _ = WEIGHTS.get(wo, True)
    qkv = K.gemm_nt(xn, wqkv_b, out_dtype=torch.bfloat16,
                    epilogue=EPI_BIAS if bqkv is not None else EPI_NONE, bias=bqkv)
    scale = dh ** -0.5
    if adrop is not None or meta.attn_bias is not None:      # weight dropout / score masks: the composed path (materialised matrix), softmax or Sinkhorn
        pscale, akeep = 1.0, None
        if adrop is not None:
            pscale, asite, pa = adrop
            akeep = draw_keep(meta, asite, (B, H, N, N), x.device, p=pa)
        o, asaved = K.attn_dropout_fwd(qkv, B, N, H, dh, scale, meta.robust, akeep, pscale, bias=meta.attn_bias)
        aux = ("attn_dropout", asaved)
    elif meta.robust:                             # robust=True: softmax + Sinkhorn normalisation (utils.py:1025-1037), fused
        p7 = {}                                   # the composed path (N > 256 / dh != 64) hands its P7 to the backward through it
        o, lse, scal = K.attn_sinkhorn_fwd(qkv, B, N, H, dh, scale, saved=p7)
        aux = (lse, scal, p7)
    else:
        o, aux = K.attn_fwd(qkv, B, N, H, dh, scale)
    if _RECORDING is not None and adrop is None and meta.attn_bias is None:
        _record(qkv, aux, B, N, H, dh, scale, meta.robust)
    keep = None
    if residual and drop is not None:
        # dropout on the branch output (vit.py:125): the bias epilogue writes the branch, one pass adds what is kept to the stream
        scale, site = drop
        keep = draw_keep(meta, site, x.shape, x.device)
        yb = K.gemm_nt(o, wo_b, out_dtype=torch.float32, epilogue=EPI_BIAS if bo is not None else EPI_NONE, bias=bo)
        y = K.dropout_add(x, yb, keep, scale, out=yb)
    elif residual:
        y = K.gemm_nt(o, wo_b, out_dtype=torch.float32, epilogue=EPI_BIAS_RESIDUAL, bias=bo, aux=x)
    else:
        y = K.gemm_nt(o, wo_b, out_dtype=torch.float32, epilogue=EPI_BIAS if bo is not None else EPI_NONE, bias=bo)
        if drop is not None:             # stand-alone module (learnable_memory_vit.py:61): dropout on the projection's output
            scale, site = drop
            keep = draw_keep(meta, site, y.shape, x.device)
            K.mask_mul_f32(y, keep, scale, out=y)
    return y, (x, xn, mean, rstd, qkv, o, aux, keep)


def attn_half_bwd(dy32: Optional[Tensor], dy16: Optional[Tensor], saved, B: int, N: int, meta: BlockMeta,
                  ln_w, ln_b, wqkv, bqkv, wo, bo, residual: bool, want_bf16: bool, want_f32: bool = True, drop_scale: float = 1.0):
    """Returns (dx32|None, dx16|None, [d ln_w, d ln_b, d wqkv, d bqkv, d wo, d bo]).

    The incoming gradient is given as fp32 (`dy32`), bf16 (`dy16`) or both; the residual add uses fp32 when present."""
    x, xn, mean, rstd, qkv, o, aux, keep = saved
    H, dh = meta.heads, meta.dim_head
    if dy16 is None:
        dy16 = K.cast_bf16(dy32)
    if keep is not None:                 # the branch sees the dropped gradient; the residual path (dres below) the whole one
        dy16 = K.mask_mul(dy16, keep, drop_scale)
    _, wo_t = WEIGHTS.get(wo, True)
    _, wqkv_t = WEIGHTS.get(wqkv, True)
    # out-proj:  y = o Wo^T (+bo) (+x)
    dwo, dbo = _dw_db(meta, dy16, o, wo, bo)
    do = K.gemm_nt(dy16, wo_t, out_dtype=torch.bfloat16)
    scale = dh ** -0.5
    if isinstance(aux, tuple) and len(aux) == 2 and isinstance(aux[0], str):         # ("attn_dropout", saved)
        dqkv = K.attn_dropout_bwd(qkv, do, aux[1], B, N, H, dh, scale)
    elif meta.robust:
        dqkv = K.attn_sinkhorn_bwd(qkv, do, aux[0], aux[1], B, N, H, dh, scale, saved=aux[2])
    else:
        dqkv = K.attn_bwd(qkv, o, do, aux, B, N, H, dh, scale)
    dwqkv, dbqkv = _dw_db(meta, dqkv, xn, wqkv, bqkv)
    if ln_w is None:                     # no LayerNorm: the projection's input gradient IS the result
        if residual:
            raise NrvError("a residual attention half needs its LayerNorm")
        dx32 = K.gemm_nt(dqkv, wqkv_t, out_dtype=torch.float32)
        return dx32, None, [None, None, dwqkv, dbqkv, dwo, dbo]
    dxn = K.gemm_nt(dqkv, wqkv_t, out_dtype=torch.bfloat16)
    tg, bg = _grad_target(meta, ln_w)
    tb, _ = _grad_target(meta, ln_b)
    if residual and keep is not None and dy32 is None:
        raise NrvError("dropout needs the fp32 residual gradient")
    dres = (dy32 if dy32 is not None else dy16) if residual else None
    dx32, dx16, dg, db = K.layernorm_bwd(dxn, x, ln_w, mean, rstd, dres=dres,
                                         want_f32=want_f32, want_bf16=want_bf16, dgamma=tg, dbeta=tb, accumulate=bg != 0.0)
    return dx32, dx16, [dg, db, dwqkv, dbqkv, dwo, dbo]


# ----------------------------------------------------------------------------------------------
# MLP half
# ----------------------------------------------------------------------------------------------
GELU_STREAM_U8 = True        # False: the bf16 gelu' stream of rounds 1 - 3 (A/B, tests)


def mlp_half_fwd(x: Tensor, meta: BlockMeta, ln_w, ln_b, w1, b1, w2, b2, residual: bool, save: bool = True, drop=None):
    """`save=False` (no gradient will be asked for): the fc1 epilogue skips the gelu'(u) output -- a 310 MB store stream per
    layer on ViT-B/16 at batch 256."""
    xn, mean, rstd = K.layernorm_fwd(x, ln_w, ln_b, meta.eps)
    w1_b, _ = WEIGHTS.get(w1, True)
    w2_b, _ = WEIGHTS.get(w2, True)
    T = x.shape[0]
    # gelu'(pre-activation) for the backward: one byte per element (include/nrv.h NRV_EPI_BIAS_GELU_Q8) unless a dropout mask is
    # going to be multiplied into it (a scaled value leaves the byte code's range)
    q8 = GELU_STREAM_U8 and drop is None and w1.shape[0] % 64 == 0
    u = torch.empty((T + 1) // 2 * 2 if q8 else T, w1.shape[0], dtype=torch.uint8 if q8 else torch.bfloat16, device=x.device) if save else None
    h = K.gemm_nt(xn, w1_b, out_dtype=torch.bfloat16, epilogue=EPI_BIAS_GELU_Q8 if q8 else EPI_BIAS_GELU, bias=b1, aux_out=u)
    keep2 = None
    if drop is not None:
        # Dropout behind the GELU (vit.py:100) and behind the second Linear (vit.py:101).  The first mask is applied to the saved
        # activation AND to the gelu' stream, so the backward (dW2 = dY^T h, dU = dY W2 o gelu') needs no mask of its own.
        scale, site = drop
        keep1 = draw_keep(meta, site, h.shape, x.device)
        K.mask_mul(h, keep1, scale, out=h)
        if u is not None:
            K.mask_mul(u, keep1, scale, out=u)
        yb = K.gemm_nt(h, w2_b, out_dtype=torch.float32, epilogue=EPI_BIAS if b2 is not None else EPI_NONE, bias=b2)
        keep2 = draw_keep(meta, site + 1, yb.shape, x.device)
        # with a residual the kept part of the branch is added to the stream; the stand-alone module (learnable_memory_vit.py:37-39) returns it
        y = K.dropout_add(x, yb, keep2, scale, out=yb) if residual else K.mask_mul_f32(yb, keep2, scale, out=yb)
    elif residual:
        y = K.gemm_nt(h, w2_b, out_dtype=torch.float32, epilogue=EPI_BIAS_RESIDUAL, bias=b2, aux=x)
    else:
        y = K.gemm_nt(h, w2_b, out_dtype=torch.float32, epilogue=EPI_BIAS if b2 is not None else EPI_NONE, bias=b2)
    return y, (x, xn, mean, rstd, u, h, keep2)


def mlp_half_bwd(dy32: Optional[Tensor], dy16: Optional[Tensor], saved, meta: BlockMeta,
                 ln_w, ln_b, w1, b1, w2, b2, residual: bool, want_bf16: bool, want_f32: bool = True, drop_scale: float = 1.0):
    x, xn, mean, rstd, u, h, keep2 = saved
    if dy16 is None:
        dy16 = K.cast_bf16(dy32)
    if keep2 is not None:
        if dy32 is None:
            raise NrvError("dropout needs the fp32 residual gradient")
        dy16 = K.mask_mul(dy16, keep2, drop_scale)
    _, w2_t = WEIGHTS.get(w2, True)
    _, w1_t = WEIGHTS.get(w1, True)
    dw2, db2 = _dw_db(meta, dy16, h, w2, b2)
    du = K.gemm_nt(dy16, w2_t, out_dtype=torch.bfloat16, epilogue=EPI_DGELU_Q8 if u.dtype == torch.uint8 else EPI_DGELU, aux=u)
    dw1, db1 = _dw_db(meta, du, xn, w1, b1)
    dxn = K.gemm_nt(du, w1_t, out_dtype=torch.bfloat16)
    tg, bg = _grad_target(meta, ln_w)
    tb, _ = _grad_target(meta, ln_b)
    dres = (dy32 if dy32 is not None else dy16) if residual else None
    dx32, dx16, dg, db = K.layernorm_bwd(dxn, x, ln_w, mean, rstd, dres=dres,
                                         want_f32=want_f32, want_bf16=want_bf16, dgamma=tg, dbeta=tb, accumulate=bg != 0.0)
    return dx32, dx16, [dg, db, dw1, db1, dw2, db2]


# ----------------------------------------------------------------------------------------------
# autograd boundary
# ----------------------------------------------------------------------------------------------
def _as_stream(x: Tensor):
    if not x.is_cuda:
        raise NrvError("noise_robust_vit_amd runs on the MI355X (HIP) device only: move the module and its input "
                       "to 'cuda'.  There is deliberately no CPU fallback on this path.")
    if x.dim() != 3:
        raise NrvError(f"expected (batch, tokens, dim), got {tuple(x.shape)}")
    B, N, D = x.shape
    x2 = x.detach().to(torch.float32).contiguous().reshape(B * N, D)
    return x2, B, N, D


def _mask_sink_grads(meta: BlockMeta, grads: List[Optional[Tensor]]) -> List[Optional[Tensor]]:
    """With a sink attached the kernels already wrote into the sink's buffers: hand autograd nothing."""
    wgrad_flush(meta)                  # autograd receives the tensors of `grads`: the queued launches must have been issued
    if meta.sink is None:
        return grads
    wgrad_join()
    return [None for _ in grads]


class EncoderStackFn(torch.autograd.Function):
    """depth x (attention half + MLP half) with residuals; params = PARAMS_PER_LAYER tensors per layer."""

    @staticmethod
    def forward(ctx, x, meta: BlockMeta, *params):
        x2, B, N, D = _as_stream(x)
        _PENDING.clear()                    # gradients queued by a backward that raised half-way never run: drop them (and their operands)
        depth = len(params) // PARAMS_PER_LAYER
        saved = []
        cur = x2
        train = any(ctx.needs_input_grad)            # inference (torch.no_grad / frozen model): keep nothing for a backward
        pdrop = float(meta.dropout)
        if not 0.0 <= pdrop < 1.0:
            raise NrvError(f"dropout probability {pdrop} outside [0, 1)")
        scale = 1.0 / (1.0 - pdrop)
        pattn = float(meta.attn_dropout)
        if not 0.0 <= pattn < 1.0:
            raise NrvError(f"attention dropout probability {pattn} outside [0, 1)")
        for i in range(depth):
            p = params[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER]
            cur, sa = attn_half_fwd(cur, B, N, meta, *p[0:6], residual=True, drop=(scale, 3 * i) if pdrop > 0.0 else None,
                                    adrop=(1.0 / (1.0 - pattn), -(2 + i), pattn) if pattn > 0.0 else None)
            cur, sm = mlp_half_fwd(cur, meta, *p[6:12], residual=True, save=train, drop=(scale, 3 * i + 1) if pdrop > 0.0 else None)
            saved.append((sa, sm) if train else None)
        ctx.meta, ctx.params, ctx.saved_blocks, ctx.shape = meta, params, saved, (B, N, D)
        ctx.drop_scale = scale
        return cur.reshape(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        meta, params, saved = ctx.meta, ctx.params, ctx.saved_blocks
        B, N, D = ctx.shape
        depth = len(params) // PARAMS_PER_LAYER
        d32 = dy.to(torch.float32).contiguous().reshape(B * N, D)
        d16 = None
        grads: List[Optional[Tensor]] = [None] * len(params)
        # the residual-stream gradient travels between the halves in fp32 (it is the sum of 2 x depth branch gradients; a bf16
        # stream was measured 1.9 % faster in round 1 and rejected: it rounds the running sum to 8 bits after every add)
        done = None                                          # layer whose weight gradients were issued, not yet declared final
        for i in reversed(range(depth)):
            p = params[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER]
            sa, sm = saved[i]
            d32, d16, gm = mlp_half_bwd(d32, d16, sm, meta, *p[6:12], residual=True, want_bf16=True, drop_scale=ctx.drop_scale)
            d32, d16, ga = attn_half_bwd(d32, d16, sa, B, N, meta, *p[0:6], residual=True, want_bf16=i > 0, drop_scale=ctx.drop_scale)
            grads[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER] = ga + gm
            # the previous layer's grouped weight-gradient launch (side stream: it ran beside this layer's dX chain) is joined
            # and its parameters declared final; then this layer's four gradients go out as one launch
            wgrad_join()
            if done is not None and meta.sink is not None:
                meta.sink.layer_done(done[0], done[1])
            wgrad_flush(meta)
            done = (i, sink_params(p))
            saved[i] = None                                  # free this block's activations early (queued operands are held by the launch)
        wgrad_join()
        if done is not None and meta.sink is not None:
            meta.sink.layer_done(done[0], done[1])
        grads = _mask_sink_grads(meta, grads)
        return (d32.reshape(B, N, D), None, *grads)


class AttnHalfFn(torch.autograd.Function):
    """Stand-alone `Attention.forward` (simple_vit.py:64-76): LN -> QKV -> attention -> out-proj, no residual."""

    @staticmethod
    def forward(ctx, x, meta: BlockMeta, ln_w, ln_b, wqkv, bqkv, wo, bo):
        x2, B, N, D = _as_stream(x)
        pa = float(meta.attn_dropout)
        pd = float(meta.dropout)
        y, saved = attn_half_fwd(x2, B, N, meta, ln_w, ln_b, wqkv, bqkv, wo, bo, residual=False,
                                 drop=(1.0 / (1.0 - pd), 0) if pd > 0.0 else None,
                                 adrop=(1.0 / (1.0 - pa), -2, pa) if pa > 0.0 else None)
        ctx.meta, ctx.params, ctx.saved_half, ctx.shape = meta, (ln_w, ln_b, wqkv, bqkv, wo, bo), saved, (B, N, D)
        ctx.drop_scale = 1.0 / (1.0 - pd) if pd > 0.0 else 1.0
        return y.reshape(B, N, wo.shape[0])

    @staticmethod
    def backward(ctx, dy):
        B, N, D = ctx.shape
        d32 = dy.to(torch.float32).contiguous().reshape(B * N, -1)
        dx32, _, g = attn_half_bwd(d32, None, ctx.saved_half, B, N, ctx.meta, *ctx.params, residual=False, want_bf16=False,
                                   drop_scale=ctx.drop_scale)
        return (dx32.reshape(B, N, D), None, *_mask_sink_grads(ctx.meta, g))


class MlpHalfFn(torch.autograd.Function):
    """Stand-alone `FeedForward.forward` (simple_vit.py:44-45): LN -> Linear -> GELU -> Linear, no residual."""

    @staticmethod
    def forward(ctx, x, meta: BlockMeta, ln_w, ln_b, w1, b1, w2, b2):
        x2, B, N, D = _as_stream(x)
        pd = float(meta.dropout)
        y, saved = mlp_half_fwd(x2, meta, ln_w, ln_b, w1, b1, w2, b2, residual=False, drop=(1.0 / (1.0 - pd), 1) if pd > 0.0 else None)
        ctx.meta, ctx.params, ctx.saved_half, ctx.shape = meta, (ln_w, ln_b, w1, b1, w2, b2), saved, (B, N, D)
        ctx.drop_scale = 1.0 / (1.0 - pd) if pd > 0.0 else 1.0
        return y.reshape(B, N, w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        B, N, D = ctx.shape
        d32 = dy.to(torch.float32).contiguous().reshape(B * N, -1)
        dx32, _, g = mlp_half_bwd(d32, None, ctx.saved_half, ctx.meta, *ctx.params, residual=False, want_bf16=False, drop_scale=ctx.drop_scale)
        return (dx32.reshape(B, N, D), None, *_mask_sink_grads(ctx.meta, g))


class PatchEmbedFn(torch.autograd.Function):
    """Patch unfold + projection + bias + positional table, written straight into the fp32 residual stream.

    SimpleViT (simple_vit.py:126-131,141-143): layout (p1 p2 c), pos = fixed sincos table [n, D], cls_slot = 0.
    VisionTransformer (vit.py:237-242,308-333,341-342, Encoder :174): layout (c p1 p2) with the Conv2d weight
    viewed [D, C*p*p]; cls_slot = 1 leaves row 0 of every sample for `class_token + pos[0]` and adds pos[1:].
    """

    @staticmethod
    def forward(ctx, img, weight, bias, pos, cls_token, patch: int, layout: int, sink):
        if not img.is_cuda:
            raise NrvError("noise_robust_vit_amd runs on the MI355X (HIP) device only: move the module and its "
                           "input to 'cuda'.  There is deliberately no CPU fallback on this path.")
        if img.requires_grad:
            raise NrvError("gradient w.r.t. the input image is not part of this hot path")
        Bn, C, H, W = img.shape
        n = (H // patch) * (W // patch)
        D = weight.shape[0]
        patches = K.patch_unfold(img.detach(), patch, layout)
        F, FP = C * patch * patch, patches.shape[1]
        if FP == F:
            wb, _ = WEIGHTS.get(weight, False)
        else:
            # patch sizes whose feature count is not a multiple of 8 (vit_h_14: 3 * 14 * 14 = 588): the unfold kernel pads the
            # rows with zero columns to the GEMM's K granularity; the projection gets matching zero columns (one small cast
            # per forward, outside the weight cache -- no BASELINE config takes this path)
            w2 = torch.nn.functional.pad(weight.detach().reshape(weight.shape[0], -1).to(torch.float32), (0, FP - F))
            wb, _ = K.cast_transpose(w2, need_t=False)
        cls_slot = 0 if cls_token is None else 1
        S = n + cls_slot
        out = torch.empty(Bn * S, D, dtype=torch.float32, device=img.device)
        pos2 = pos.detach().reshape(-1, D).to(torch.float32)
        if pos2.shape[0] != S:
            # the GEMM epilogue reads the table through raw pointers (row = token index): a table of another grid would be
            # read out of bounds or add the wrong positions (the reference's broadcast add raises for a learned table)
            raise NrvError(f"positional table has {pos2.shape[0]} rows but the image gives {n} patches"
                           f"{' + class token' if cls_slot else ''} = {S} tokens")
        if C * patch * patch != weight[0].numel():
            raise NrvError(f"patch features {C * patch * patch} != projection in_features {weight[0].numel()}")
        if cls_slot:
            K.gemm_nt(patches, wb, epilogue=EPI_BIAS_RESIDUAL, bias=bias.detach() if bias is not None else None,
                      aux=pos2[1:], aux_row_mod=n, out=out, out_group=n, out_group_stride=S, out_row_offset=1)
            out.view(Bn, S, D)[:, 0] = (cls_token.detach().reshape(D) + pos2[0])      # B*D elements: not the hot path
        else:
            K.gemm_nt(patches, wb, epilogue=EPI_BIAS_RESIDUAL, bias=bias.detach() if bias is not None else None,
                      aux=pos2, aux_row_mod=n, out=out)
        ctx.save_for_backward(patches)
        ctx.cfg = (Bn, n, S, D, cls_slot, weight.shape, bias is not None, pos.shape, pos.requires_grad,
                   cls_token is not None and cls_token.requires_grad, sink, weight, bias)
        return out.reshape(Bn, S, D)

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        Bn, n, S, D, cls_slot, wshape, has_bias, pos_shape, pos_grad, cls_grad, sink, weight, bias = ctx.cfg
        d32 = dy.to(torch.float32).contiguous().reshape(Bn * S, D)
        d16 = K.cast_bf16(d32)
        tw, bw = (sink.target(weight) if sink is not None else (None, 0.0))
        if tw is not None:
            tw = tw.reshape(D, -1)
        F = weight[0].numel()
        padded = patches.shape[1] != F               # zero feature columns (patch size 14): gradient of the real columns only
        tgt = None if padded else tw
        if cls_slot:
            dw = K.gemm_tn(d16, patches, out=tgt, beta=0.0 if padded else bw, a_group=n, a_group_stride=S, a_row_offset=1, T=Bn * n)
        else:
            dw = K.gemm_tn(d16, patches, out=tgt, beta=0.0 if padded else bw)
        if padded:
            dw = dw[:, :F]
            if tw is not None:
                tw.copy_(dw) if bw == 0.0 else tw.add_(dw)
        # positional-table / class-token / bias gradients are reductions of dy over the batch
        dpos = dcls = db = None
        if pos_grad or cls_grad or cls_slot:
            dsum = K.colsum(d16.reshape(Bn, S * D)).reshape(S, D)
            dpos = dsum.reshape(pos_shape) if pos_grad else None
            dcls = dsum[0].reshape(1, 1, D) if cls_grad else None
            if has_bias:
                db = dsum[cls_slot:].sum(0)          # [S, D] -> [D]: outside the hot path
        elif has_bias:
            db = K.colsum(d16)
        if has_bias and sink is not None:
            tb, bb = sink.target(bias)
            tb.copy_(db) if bb == 0.0 else tb.add_(db)
        if sink is not None:
            sink.layer_done(-1, [weight] + ([bias] if has_bias else []))
            return (None, None, None, dpos, dcls, None, None, None)
        return (None, dw.reshape(wshape), db, dpos, dcls, None, None, None)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b through the MFMA GEMM (bf16 operands, fp32 result) for projections next to the encoder
    (MAE `enc_to_dec` / `to_pixels`, mae.py:34-49).  x [..., K] fp32, K % 8 == 0, out_features % 8 == 0."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if not x.is_cuda:
            raise NrvError("noise_robust_vit_amd runs on the MI355X (HIP) device only; there is no CPU fallback")
        shp = x.shape
        x2 = x.detach().to(torch.float32).contiguous().reshape(-1, shp[-1])
        xb = K.cast_bf16(x2)
        wb, _ = WEIGHTS.get(weight, True)
        y = K.gemm_nt(xb, wb, out_dtype=torch.float32, epilogue=EPI_BIAS if bias is not None else EPI_NONE,
                      bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(xb)
        ctx.meta = (weight, bias is not None, shp)
        return y.reshape(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (xb,) = ctx.saved_tensors
        weight, has_bias, shp = ctx.meta
        d16 = K.cast_bf16(dy.to(torch.float32).contiguous().reshape(-1, weight.shape[0]))
        _, wt = WEIGHTS.get(weight, True)
        if has_bias:
            dw, db = K.gemm_tn(d16, xb, want_dbias=True)
        else:
            dw, db = K.gemm_tn(d16, xb), None
        dx = K.gemm_nt(d16, wt, out_dtype=torch.float32)
        return dx.reshape(shp), dw, db


def _flat_token_index(index: Tensor, n: int) -> Tensor:
    B = index.shape[0]
    return (index + torch.arange(B, device=index.device)[:, None] * n).reshape(-1).contiguous()


class GatherTokensFn(torch.autograd.Function):
    """tokens[b, index[b, :]] (MAE keeps a random 25 % of the tokens, mae.py:75-76; reads the decoder's output at the masked
    positions, mae.py:112); backward scatters rows back.  Bounds-checked row kernels, no library sort in either direction."""

    @staticmethod
    def forward(ctx, tokens, index):
        B, n, D = tokens.shape
        flat = _flat_token_index(index, n)
        out = K.gather_rows(tokens.detach().to(torch.float32).contiguous().reshape(B * n, D), flat)
        ctx.save_for_backward(flat)
        ctx.shape = (B, n, D, index.shape[1])
        return out.reshape(B, index.shape[1], D)

    @staticmethod
    def backward(ctx, dy):
        (flat,) = ctx.saved_tensors
        B, n, D, k = ctx.shape
        d = K.scatter_rows(dy.to(torch.float32).contiguous().reshape(B * k, D), flat, B * n)
        return d.reshape(B, n, D), None


class ScatterTokensFn(torch.autograd.Function):
    """out[b, index[b, j]] = rows[b, j], zeros elsewhere (MAE re-assembles the full-length decoder sequence from the kept tokens,
    mae.py:103-107; indices unique per sample); backward gathers.  The counterpart of GatherTokensFn: torch's index_put / Embedding
    backward would bring library sort kernels into the step (and into its HIP graph)."""

    @staticmethod
    def forward(ctx, rows, index, n: int):
        B, k, D = rows.shape
        flat = _flat_token_index(index, n)
        out = K.scatter_rows(rows.detach().to(torch.float32).contiguous().reshape(B * k, D), flat, B * n)
        ctx.save_for_backward(flat)
        ctx.shape = (B, n, D, k)
        return out.reshape(B, n, D)

    @staticmethod
    def backward(ctx, dy):
        (flat,) = ctx.saved_tensors
        B, n, D, k = ctx.shape
        d = K.gather_rows(dy.to(torch.float32).contiguous().reshape(B * n, D), flat)
        return d.reshape(B, k, D), None, None


def flat_layer_params(layers: Sequence[Sequence[Optional[Tensor]]]) -> List[Optional[Tensor]]:
    out: List[Optional[Tensor]] = []
    for lp in layers:
        if len(lp) != PARAMS_PER_LAYER:
            raise ValueError("each layer needs PARAMS_PER_LAYER entries")
        out.extend(lp)
    return out
