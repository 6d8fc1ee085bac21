"""One fused cast of all autocast-consumed parameters per forward.

Under autocast every Conv2d / Linear casts its fp32 weight (and bias) to bf16/fp16 with its own kernel and its own
autograd node, and the backward casts every weight gradient back with another kernel: ~200 + ~200 launches per step for
ResNet-101 + the transformer -- on MI355X the training step is host/launch-bound, so they matter.
Here the model casts ALL such parameters once per forward with ONE launch (`FusedCast`: csrc/multi_cast.hip writes every
working copy into a persistent flat buffer; the backward casts every gradient back with one launch), and the layers below
pick the low-precision copies up through `lookup`.  Outside autocast nothing changes (the table is empty and `lookup`
returns the parameter itself).  Numerically identical to autocast's own per-op casts.

The module also hosts the GEMM-shaped layer variants that ride on those copies: 1x1 convolutions of channels-last maps as
hipBLASLt GEMMs, weight gradients over many rows as row-split batched GEMMs (`weight_grad`), and `linear` for token
matrices.
"""
import contextlib
import os

import torch
import torch.nn.functional as F
from torch import nn

from .ops.functions import conv_gemm_func

_ACTIVE = {}        # id(fp32 parameter) -> low-precision copy, valid inside one `scope`
ENABLED = True      # A/B switch (tests compare against autocast's own per-op casts)
GEMM_1X1 = os.environ.get("OCPG_GEMM_1X1", "1") != "0"     # A/B switch: 1x1 convs of channels-last maps as hipBLASLt GEMMs instead of MIOpen convolutions
GEMM_3X3 = os.environ.get("OCPG_GEMM_3X3", "0") != "0"     # A/B switch: 3x3 convs of channels-last maps as HIP im2col + one hipBLASLt GEMM
SPLITK_3X3 = os.environ.get("OCPG_SPLITK_3X3", "1") != "0"     # A/B switch: 3x3 convs with few output pixels and a long reduction through the split-K MFMA kernel (0 = MIOpen)
SPLIT_K = os.environ.get("OCPG_SPLIT_K", "1") != "0"      # A/B switch: weight gradients over many rows as row-split batched GEMMs


MULTI_CAST = True   # A/B switch: one HIP launch for all parameter casts / gradient casts (csrc/multi_cast.hip)

# ---- weight gradients on a side stream --------------------------------------------------------------------------------------
# In the backward of a layer only the INPUT gradient is on the critical path; the weight gradient is first read by
# FusedCast.backward, the very last node of the backward pass.  The fused conv nodes of the ResNet body therefore compute it on
# a second stream: in eager mode the two streams overlap, and a captured step gets PARALLEL BRANCHES in its HIP graph instead
# of one chain (most of these kernels do not fill 256 CUs, and a chain pays every launch gap).  Only for weights that are
# FusedCast copies used ONCE per forward (the engine would otherwise sum two gradients on the main stream before the join).
WGRAD_STREAM = os.environ.get("OCPG_WGRAD_STREAM", "0") != "0"     # measured (round 3, graph mode): 42.1 ms with, 40.3 ms without -- the HIP graph executor does not overlap the branches; kept as an A/B switch
_side_streams = {}
_side_pending = {}


class side_wgrad:
    """`with side_wgrad(flag, *inputs) as sw: ...` runs the body on the device's weight-gradient stream when `flag` (the weight is
    a FusedCast copy: is_cast_copy, decided in the node's forward); the inputs are marked as used there (the allocator must not hand their memory out again before that stream is done);
    outputs allocated inside belong to the side stream and are marked as used on the main stream by `publish`."""

    def __init__(self, on, *inputs):
        self.on = bool(on and WGRAD_STREAM and inputs[0].is_cuda)
        self.inputs = inputs

    def __enter__(self):
        if not self.on:
            return self
        self.main = torch.cuda.current_stream()
        dev = self.main.device
        side = _side_streams.get(dev)
        if side is None:
            side = _side_streams[dev] = torch.cuda.Stream(device=dev)
        self.side = side
        side.wait_stream(self.main)
        for t in self.inputs:
            t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        _side_pending[dev] = True
        return self

    def publish(self, t):
        if self.on:
            t.record_stream(self.main)
        return t

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False


# ---- row-split weight-gradient partials summed inside the fused gradient cast ------------------------------------------------------
DEFER_SUM = os.environ.get("OCPG_DEFER_WGRAD_SUM", "1") != "0"     # A/B switch
_PARTIALS = {}          # data_ptr of slice 0 -> (splits, stride in elements, numel of a slice)


def defer_sum(part, as_dtype=None):
    """part [S, ...] partial products of ONE weight gradient -> slice 0, with the promise that FusedCast.backward adds the other S-1
    slices while it casts (csrc/multi_cast.hip: multi_cast_sum).  Only for the working copy of a parameter that is used ONCE per
    forward: the autograd engine must hand this very tensor to FusedCast.backward (it would sum two uses into a new tensor; an
    entry that nobody consumed raises at the next forward).  as_dtype: the partials are fp32 but the gradient autograd expects has
    this (low-precision) dtype: slice 0 is handed over as a low-precision VIEW of the same memory (never read as such)."""
    s0 = part[0]
    if as_dtype is not None and as_dtype != part.dtype:
        assert part.dtype == torch.float32 and part.is_contiguous()
        _PARTIALS[part.data_ptr()] = (part.shape[0] | (1 << 40), part.stride(0), s0.numel())
        return part.view(as_dtype)[0, :s0.numel()].view(s0.shape)       # same address, the dtype the engine checks
    _PARTIALS[part.data_ptr()] = (part.shape[0], part.stride(0), s0.numel())
    return s0


def mark_single_use(*modules_or_params):
    """Declare that these parameters are used ONCE per forward, so the sums over the row-split partials of their gradients (and the
    column sums behind bias gradients) may be left to the fused gradient cast (defer_sum).  NOT for shared modules (a module
    applied to several levels: the engine sums the uses into a new tensor and the partials would be lost -- caught at the next
    forward, loudly)."""
    for m in modules_or_params:
        for p_ in (m.parameters() if isinstance(m, nn.Module) else [m]):
            p_._ocpg_single_use = True


def deferrable(w):
    """True for a FusedCast working copy of a parameter marked single-use (tagged by `scope`)."""
    return DEFER_SUM and MULTI_CAST and w is not None and getattr(w, "_ocpg_defer", False)


def colsum(gy2, like):
    """gy2 [R, C].sum(0) as the gradient of the bias working copy `like`; deferred (one launch of partials now, the sum inside the
    fused gradient cast) when `like` is deferrable and gy2 is a low-precision GPU matrix."""
    if deferrable(like) and gy2.is_cuda and gy2.dtype in (torch.bfloat16, torch.float16) and gy2.is_contiguous() and like.dtype == gy2.dtype:
        from .._lib import check, lib
        r, c = gy2.shape
        nb = int(lib().ocpg_colsum_blocks(r))
        part = torch.empty((nb, c), dtype=torch.float32, device=gy2.device)
        check(lib().ocpg_colsum_partials(gy2.data_ptr(), r, c, _DT[gy2.dtype], part.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_colsum_partials")
        return defer_sum(part, gy2.dtype)
    if gy2.is_cuda and gy2.dtype == torch.float32 and gy2.is_contiguous() and gy2.shape[0] >= 4096:
        # fp32 islands (the MSDeformAttn projections over all 51 000 tokens): per-block column sums in ONE streaming launch (12 us for
        # a 52-MB gradient) + a sum over ~400 partial rows, instead of ATen's column reduction of the whole matrix (34 us)
        from .._lib import check, lib
        r, c = gy2.shape
        nb = int(lib().ocpg_colsum_blocks(r))
        part = torch.empty((nb, c), dtype=torch.float32, device=gy2.device)
        check(lib().ocpg_colsum_partials(gy2.data_ptr(), r, c, 0, part.data_ptr(), torch.cuda.current_stream().cuda_stream), "ocpg_colsum_partials")
        return part.sum(0)
    return gy2.sum(0)


def _finish_partials(g):
    """g as it arrived in FusedCast.backward; when it is slice 0 of a deferred partial sum (defer_sum), the sum over all slices."""
    pr = _PARTIALS.pop(g.data_ptr(), None) if _PARTIALS else None
    if pr is None:
        return g
    splits, stride, numel = pr[0] & ((1 << 40) - 1), pr[1], pr[2]
    dt = torch.float32 if pr[0] >> 40 else g.dtype            # flagged: fp32 partials behind a low-precision view of slice 0
    esz = torch.empty((), dtype=dt).element_size()
    part = torch.empty(0, dtype=dt, device=g.device).set_(g.untyped_storage(), g.storage_offset() * g.element_size() // esz,
                                                          (splits, numel), (stride, 1))
    if g.is_contiguous():
        return part.sum(0).view(g.shape)
    # slice 0 keeps the memory order of the layout the layer computed it in (dense, e.g. channels-last): sum in that order
    out = torch.empty_like(g, dtype=dt)
    out.as_strided((numel,), (1,), out.storage_offset()).copy_(part.sum(0))
    return out


def is_cast_copy(w):
    """True for the low-precision working copy FusedCast made of a parameter (its gradient is first read by FusedCast.backward)."""
    return w.grad_fn is not None and w.grad_fn.name() == "FusedCastBackward"


def join_wgrad():
    """The main stream waits for every weight gradient computed on the side stream (FusedCast.backward; end of a backward pass)."""
    if not _side_pending:
        return
    cur = torch.cuda.current_stream()
    if _side_pending.pop(cur.device, False):
        cur.wait_stream(_side_streams[cur.device])

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_CHUNK = 2048
_BIAS_CHUNK = 256


def _dense(t):
    """Non-overlapping and dense: an elementwise copy to a tensor with the same strides keeps the memory order."""
    if t.is_contiguous():
        return True
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def _same_layout(g, shape, strides):
    return tuple(g.shape) == tuple(shape) and all(a == b or n == 1 for a, b, n in zip(g.stride(), strides, shape))


class _CastPlan:
    """Static part of the fused cast of one parameter set on one device: flat offsets (16-byte aligned), chunk prefix."""

    def __init__(self, params, low_dtype):
        self.low_dtype = low_dtype
        self.shapes = [tuple(p.shape) for p in params]
        self.strides = [p.stride() for p in params]
        self.numels = [p.numel() for p in params]
        self.offsets, off = [], 0
        for n in self.numels:
            self.offsets.append(off)
            off += (n + 7) // 8 * 8
        self.total = off
        self.device = params[0].device
        prefix = [0]
        for n in self.numels:
            prefix.append(prefix[-1] + (n + _CHUNK - 1) // _CHUNK)
        self.chunks = prefix[-1]
        self.meta = torch.tensor([self.numels + [0], prefix], dtype=torch.int64).to(self.device)      # [2, n+1]
        # persistent working copies: the weights only change at optimizer.step, so re-writing the buffer in the next forward
        # (through the raw pointer: no autograd version bump) can only ever store identical values under a live graph
        self.low = torch.empty(self.total, dtype=low_dtype, device=self.device)
        self.low_ptrs = torch.tensor([self.low.data_ptr() + o * self.low.element_size() for o in self.offsets], dtype=torch.int64).to(self.device)
        self.src_ptrs_host = [p.data_ptr() for p in params]
        self.src_ptrs = torch.tensor(self.src_ptrs_host, dtype=torch.int64).to(self.device)

    def cast_params(self):
        from .._lib import check, lib
        check(lib().ocpg_multi_cast(self.src_ptrs.data_ptr(), self.low_ptrs.data_ptr(), self.meta[0].data_ptr(), self.meta[1].data_ptr(),
                                    len(self.numels), self.chunks, 0, _DT[self.low_dtype], torch.cuda.current_stream().cuda_stream),
              "ocpg_multi_cast")
        low = self.low      # fresh view objects per forward (each forward's outputs carry their own autograd history)
        return [low.as_strided(sh, st, o) for sh, st, o in zip(self.shapes, self.strides, self.offsets)]

    def cast_grads(self, grads, idx):
        """grads[i] (low precision, same layout as parameter i) for i in idx -> fp32 views of ONE fresh flat buffer.
        The per-call pointer table goes up through a small ring of PINNED staging buffers (a pageable host->device copy
        would stall the host until the stream drains, measured +7 ms per step)."""
        from .._lib import check, lib
        n = len(idx)
        flat = torch.empty(self.total, dtype=torch.float32, device=self.device)
        base = flat.data_ptr()
        parts = [_PARTIALS.pop(grads[i].data_ptr(), None) if _PARTIALS else None for i in idx]
        prefix = [0]
        for i, pr in zip(idx, parts):
            # fp32 bias-gradient partials (flag bit 40: few elements, hundreds of slices) are folded by a workgroup per 256 elements
            # (csrc/multi_cast.hip kBiasChunk): one workgroup per 2 048 took ~1 ms for the encoder FFNs' 2 048-wide biases
            ch = _BIAS_CHUNK if (pr is not None and pr[0] >> 40) else _CHUNK
            prefix.append(prefix[-1] + (self.numels[i] + ch - 1) // ch)
        for i, pr in zip(idx, parts):
            if pr is not None and pr[2] != self.numels[i]:
                raise RuntimeError(f"deferred weight-gradient partials do not match the parameter they arrived for: parameter {i} of shape "
                                   f"{self.shapes[i]}, gradient {tuple(grads[i].shape)}, partials {pr}")
        any_sum = any(pr is not None for pr in parts)
        table = torch.tensor([[grads[i].data_ptr() for i in idx], [base + 4 * self.offsets[i] for i in idx],
                              [self.numels[i] for i in idx], prefix[:-1],
                              [1 if pr is None else pr[0] for pr in parts], [0 if pr is None else pr[1] for pr in parts]], dtype=torch.int64)
        if torch.cuda.is_current_stream_capturing():
            # a captured host->device copy re-reads its HOST source on every replay: give this capture its own pinned table,
            # alive as long as the plan (the ring below is overwritten by later eager steps); the pointers it holds are the
            # graph's private-pool addresses, identical on every replay
            spares = self.__dict__.get("_capture_spares")
            if not spares:
                raise RuntimeError("fused gradient cast: run one eager backward before capturing (pinned tables are allocated there)")
            host = spares.pop()                                  # pinned memory cannot be allocated while capturing
            dev = torch.empty((6, len(self.numels)), dtype=torch.int64, device=self.device)
            self.__dict__.setdefault("_capture_tables", []).append((host, dev))
            host[:, :n] = table
            dev.copy_(host, non_blocking=True)
        else:
            ring = self.__dict__.get("_ring")
            if ring is None:
                cap = len(self.numels)
                ring = self._ring = {"host": [torch.empty((6, cap), dtype=torch.int64).pin_memory() for _ in range(8)],
                                     "dev": [torch.empty((6, cap), dtype=torch.int64, device=self.device) for _ in range(8)],
                                     "event": [None] * 8, "next": 0}
                self._capture_spares = [torch.empty((6, cap), dtype=torch.int64).pin_memory() for _ in range(4)]
            k = ring["next"]
            ring["next"] = (k + 1) % 8
            if ring["event"][k] is not None:
                ring["event"][k].synchronize()          # slot still in flight only if the host is 8 backward passes ahead
            host, dev = ring["host"][k], ring["dev"][k]
            host[:, :n] = table
            dev.copy_(host, non_blocking=True)
            ev = ring["event"][k] = torch.cuda.Event()
            ev.record()
        # the last prefix entry (the total) is passed by value; the kernel's search never reads chunk_prefix[n]
        if any_sum:
            check(lib().ocpg_multi_cast_sum(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(),
                                            dev[5].data_ptr(), n, prefix[-1], _DT[self.low_dtype], 0, torch.cuda.current_stream().cuda_stream),
                  "ocpg_multi_cast_sum")
        else:
            check(lib().ocpg_multi_cast(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), n, prefix[-1],
                                        _DT[self.low_dtype], 0, torch.cuda.current_stream().cuda_stream), "ocpg_multi_cast")
        return [flat.as_strided(self.shapes[i], self.strides[i], self.offsets[i]) for i in idx]


class _NoPlan:
    """Remembers that this parameter set cannot take the one-launch path (non-dense or non-fp32 members)."""

    def __init__(self, ptrs):
        self.src_ptrs_host = ptrs


class FusedCast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dtype, plan, *params):
        ctx.src_dtype = params[0].dtype
        ctx.src_strides = [p.stride() for p in params]
        ctx.plan = plan
        if plan is not None:
            outs = tuple(plan.cast_params())
        else:
            outs = [torch.empty_like(p, dtype=dtype) for p in params]
            torch._foreach_copy_(outs, list(params))
            outs = tuple(outs)
        # A custom Function marks EVERY output as requiring grad when ANY input does.  The copies of frozen parameters (the
        # ResNet stem and layer1, models/backbone.py:63-65) must stay non-differentiable, or autograd runs the backward of the
        # frozen layers too (dgrad + wgrad of the three largest bottlenecks and the stem, then throws the results away).
        frozen = [o for o, p in zip(outs, params) if not p.requires_grad]
        if frozen:
            ctx.mark_non_differentiable(*frozen)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        join_wgrad()            # weight gradients computed on the side stream are first read here
        idx = [i for i, g in enumerate(grads) if g is not None and ctx.needs_input_grad[i + 2]]
        res = [None] * len(grads)
        if idx:
            plan = ctx.plan
            fast = [i for i in idx if plan is not None and grads[i].dtype == plan.low_dtype and _same_layout(grads[i], plan.shapes[i], plan.strides[i])]
            if fast:
                for i, o in zip(fast, plan.cast_grads(grads, fast)):
                    res[i] = o
            slow = [i for i in idx if res[i] is None]
            if slow:
                # gradients take the PARAMETER's strides (DDP's gradient-as-bucket-view layout contract: a channels-last
                # 1x1 weight [Co,Ci,1,1] and its dense gradient share the memory order but not the nominal strides)
                outs = [torch.empty_strided(grads[i].shape, ctx.src_strides[i], dtype=ctx.src_dtype, device=grads[i].device) for i in slow]
                # a deferred partial sum that arrives HERE (its layout does not match the plan's, e.g. a weight that is not
                # channels-last under channels-last activations) is finished now: slice 0 alone would be 1/splits of the gradient
                torch._foreach_copy_(outs, [_finish_partials(grads[i]) for i in slow])
                for i, o in zip(slow, outs):
                    res[i] = o
        return (None, None, *res)


def lookup(p):
    """The low-precision copy of parameter `p` for the current forward, or `p` itself."""
    return _ACTIVE.get(id(p), p) if _ACTIVE else p


def _split_rows(m):
    """Number of row chunks for the weight-gradient contraction over m rows (a divisor of m near m/1536, or 1)."""
    s = _SPLITS.get(m)
    if s is None:
        s = 1
        if m >= 4096:
            want = m / _CHUNK_ROWS
            cands = [d for d in range(max(2, int(m // (2 * _CHUNK_ROWS))), min(128, int(m // (_CHUNK_ROWS / 2))) + 1) if m % d == 0]
            if cands:
                s = min(cands, key=lambda d: abs(d - want))
        _SPLITS[m] = s
    return s


_SPLITS = {}
_CHUNK_ROWS = float(os.environ.get("OCPG_SPLIT_ROWS", "1536"))      # rows per chunk the split aims for


def _mm(a, b, trans_b=False, bias=None):
    from .ops.functions.gemm_func import mm
    return mm(a, b, trans_b, bias)


def weight_grad(gy2, x2, defer=False):
    """gy2 [M, Co], x2 [M, Ci] (row-major, M = pixels or tokens, large) -> gy2^T x2 [Co, Ci].  defer: leave the sum over the row
    chunks to the fused gradient cast (defer_sum; the caller checked `deferrable(weight copy)`).

    As ONE GEMM this has a tiny output and a huge reduction dimension: hipBLASLt runs it on a handful of workgroups
    (measured on MI355X, bf16: M=38400, 512x128 -> 138 us; M=153600, 256x64 -> 360 us; fp32 M=51000, 256x256 -> 201 us).
    Split the rows into S chunks -> a batched GEMM with S x more workgroups + one small reduction: 25-36 us
    (tools/bench_conv1x1_bwd.py)."""
    m = gy2.shape[0]
    s = _split_rows(m) if SPLIT_K else 1
    if gy2.is_cuda:
        from .ops.functions.gemm_func import mm_tn
        return mm_tn(gy2, x2, s, defer and gy2.dtype in (torch.bfloat16, torch.float16))
    if s == 1:
        return torch.mm(gy2.t(), x2)
    return torch.bmm(gy2.view(s, m // s, -1).transpose(1, 2), x2.view(s, m // s, -1)).sum(0)


class Conv1x1AsGemm(torch.autograd.Function):
    """A 1x1 / stride-1 convolution of a channels-last map IS the GEMM [N*H*W, Cin] x [Cin, Cout] on views (no copies).
    Measured on MI355X at the ResNet-101 shapes of config #2 (10 frames, bf16, GPU-busy time, tools/bench_conv1x1*.py):
    forward 14-18 us against MIOpen's 58-125 us (layers 3/4), input gradient 14-29 us against 26-95 us, weight gradient
    (split over the rows, `weight_grad`) 23-36 us against 61-81 us.  One autograd node."""

    @staticmethod
    def forward(ctx, x, w, bias):
        n, c, h, wd = x.shape
        x2 = x.permute(0, 2, 3, 1).reshape(n * h * wd, c)
        w2 = w.reshape(w.shape[0], c)
        y2 = _mm(x2, w2, True, bias)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.defer_w, ctx.defer_b = deferrable(w), bias if deferrable(bias) else None
        return y2.view(n, h, wd, w.shape[0]).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        n, c, h, wd = x.shape
        co = w.shape[0]
        gy2 = gy.permute(0, 2, 3, 1).reshape(n * h * wd, co)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _mm(gy2, w.reshape(co, c)).view(n, h, wd, c).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = weight_grad(gy2, x.permute(0, 2, 3, 1).reshape(n * h * wd, c), ctx.defer_w).view(w.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = colsum(gy2, ctx.defer_b)
        return gx, gw, gb


class TokenLinearFunction(torch.autograd.Function):
    """y = x2 W^T + b for MANY rows (tokens): same GEMMs as F.linear, but the weight gradient goes through `weight_grad`."""

    @staticmethod
    def forward(ctx, x2, w, bias):
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.defer_w, ctx.defer_b = deferrable(w), bias if deferrable(bias) else None
        return _mm(x2, w, True, bias)

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _mm(gy, w) if ctx.needs_input_grad[0] else None
        gw = weight_grad(gy, x2, ctx.defer_w) if ctx.needs_input_grad[1] else None
        gb = colsum(gy, ctx.defer_b) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


TOKEN_LINEAR_MIN_ROWS = int(os.environ.get("OCPG_TOKEN_LINEAR_MIN_ROWS", "8192"))     # rows from which a Linear goes through TokenLinearFunction (plan-cache GEMMs, row-split weight gradient)
SMALL_LINEAR_F32 = os.environ.get("OCPG_SMALL_LINEAR_F32", "1") != "0"     # A/B switch: the fp32 islands' few-row Linears too (csrc/small_linear_f32.hip)
SMALL_LINEAR = os.environ.get("OCPG_SMALL_LINEAR", "1") != "0"     # A/B switch: few-row Linears as one launch each way (csrc/small_linear.hip)


class SmallLinearFunction(torch.autograd.Function):
    """y = x W^T + b for FEW rows under bf16 autocast: one launch forward (input cast, GEMM, bias), ONE launch backward (input
    gradient in x's dtype, weight gradient, bias gradient) instead of cast + addmm and mm + mm + sum + cast."""

    @staticmethod
    def forward(ctx, x, w, b, relu=False):
        from .._lib import check, lib
        k, co = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y = torch.empty((*x.shape[:-1], co), dtype=torch.bfloat16, device=x.device)    # not a view: an in-place ReLU may follow
        check(lib().ocpg_small_linear_fwd(x2.data_ptr(), int(x2.dtype == torch.float32), w.data_ptr(), None if b is None else b.data_ptr(),
                                          x2.shape[0], k, co, int(relu), y.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_small_linear_fwd")
        if relu:
            ctx.save_for_backward(x2, w, y)
        else:
            ctx.save_for_backward(x2, w)
        ctx.x_shape, ctx.has_bias, ctx.relu = x.shape, b is not None, bool(relu)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .._lib import check, lib
        x2, w = ctx.saved_tensors[:2]
        ymask = ctx.saved_tensors[2] if ctx.relu else None
        co, k = w.shape
        g2 = gy.reshape(-1, co)
        if g2.dtype not in (torch.float32, torch.bfloat16):
            g2 = g2.float()
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(co, dtype=torch.bfloat16, device=w.device) if ctx.has_bias else None
        check(lib().ocpg_small_linear_bwd(g2.data_ptr(), int(g2.dtype == torch.float32), x2.data_ptr(), int(x2.dtype == torch.float32), w.data_ptr(),
                                          None if ymask is None else ymask.data_ptr(), x2.shape[0], k, co,
                                          None if gx is None else gx.data_ptr(), gw.data_ptr(), None if gb is None else gb.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "ocpg_small_linear_bwd")
        return (None if gx is None else gx.view(ctx.x_shape)), gw, gb, None


class SmallLinearF32Function(torch.autograd.Function):
    """y = x W^T + b for FEW rows in fp32 (outside autocast: MSDeformAttn's projections over the decoder's query rows): one launch
    forward, one backward (csrc/small_linear_f32.hip) instead of addmm and mm + mm + sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        from .._lib import check, lib
        k, co = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y = torch.empty((*x.shape[:-1], co), dtype=torch.float32, device=x.device)
        check(lib().ocpg_small_linear_f32_fwd(x2.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), x2.shape[0], k, co, y.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "ocpg_small_linear_f32_fwd")
        ctx.save_for_backward(x2, w)
        ctx.x_shape, ctx.has_bias = x.shape, b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        from .._lib import check, lib
        x2, w = ctx.saved_tensors
        co, k = w.shape
        g2 = gy.reshape(-1, co)
        if g2.dtype != torch.float32:
            g2 = g2.float()
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(co, dtype=torch.float32, device=w.device) if ctx.has_bias else None
        check(lib().ocpg_small_linear_f32_bwd(g2.data_ptr(), x2.data_ptr(), w.data_ptr(), x2.shape[0], k, co, None if gx is None else gx.data_ptr(),
                                              gw.data_ptr(), None if gb is None else gb.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_small_linear_f32_bwd")
        return (None if gx is None else gx.view(ctx.x_shape)), gw, gb


def _small_linear_f32_ok(x, w, b):
    k = x.shape[-1]
    return (SMALL_LINEAR_F32 and x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled("cuda") and w.dim() == 2 and k % 64 == 0
            and k <= 512 and 0 < x.numel() <= 1024 * k and x.dtype == torch.float32 and w.dtype == torch.float32 and w.is_contiguous()
            and w.requires_grad and (b is None or (b.dtype == torch.float32 and b.is_contiguous())))


def _small_linear_ok(x, w, b):
    k = x.shape[-1]
    return (SMALL_LINEAR and x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and w.dim() == 2 and k % 64 == 0 and k <= int(os.environ.get("OCPG_SMALL_LINEAR_MAXK", "512"))
            and x.numel() <= 1024 * k and x.dtype in (torch.float32, torch.bfloat16) and w.dtype == torch.bfloat16 and w.is_contiguous()
            and (b is None or (b.dtype == torch.bfloat16 and b.is_contiguous())))


def linear_relu(x, w, b):
    """relu(linear(x, w, b)); few rows under bf16 autocast: ONE launch (the ReLU rides in the GEMM epilogue, its mask in the backward's loads)."""
    if _small_linear_ok(x, w, b):
        return SmallLinearFunction.apply(x, w, b, True)
    return F.relu(linear(x, w, b))


def linear(x, w, b):
    """F.linear; on the GPU, with >= 8192 rows and a weight that needs a gradient, through TokenLinearFunction; with few rows under
    bf16 autocast through SmallLinearFunction."""
    k = x.shape[-1]
    if _small_linear_ok(x, w, b):
        return SmallLinearFunction.apply(x, w, b)
    if _small_linear_f32_ok(x, w, b):
        return SmallLinearF32Function.apply(x, w, b)
    if SPLIT_K and x.is_cuda and w.requires_grad and x.numel() >= TOKEN_LINEAR_MIN_ROWS * k and torch.is_grad_enabled():
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x, w, b = x.to(dt), w.to(dt), None if b is None else b.to(dt)
        if x.dtype == w.dtype:
            return TokenLinearFunction.apply(x.reshape(-1, k), w, b).view(*x.shape[:-1], w.shape[0])
    return F.linear(x, w, b)


class TokenLinear(nn.Linear):
    """nn.Linear for modules that run OUTSIDE autocast (fp32 islands: MSDeformAttn's projections over all H*W tokens)."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class Conv2d(nn.Conv2d):
    def _is_pointwise(self):
        r = self.__dict__.get("_pointwise")
        if r is None:
            r = self.__dict__["_pointwise"] = (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
                                               and self.groups == 1 and self.padding_mode == "zeros")
        return r

    def forward(self, x):
        w, b = lookup(self.weight), None if self.bias is None else lookup(self.bias)
        if GEMM_1X1 and x.is_cuda and x.dim() == 4 and self._is_pointwise() and x.is_contiguous(memory_format=torch.channels_last):
            if torch.is_autocast_enabled("cuda"):
                dt = torch.get_autocast_dtype("cuda")
                x, w, b = x.to(dt), w.to(dt), None if b is None else b.to(dt)
            if x.dtype == w.dtype:
                return Conv1x1AsGemm.apply(x, w, b)
        elif SPLITK_3X3 and x.is_cuda and x.dim() == 4 and self.kernel_size == (3, 3) and torch.is_autocast_enabled("cuda") \
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16:
            from .ops.functions import conv_bn_func
            if conv_bn_func.eligible3x3_splitk(x, self):       # few output pixels, long reduction: the neck's stride-2 level (csrc/conv3x3_mfma.hip, split-K)
                return conv_bn_func.conv3x3_splitk(x, w, b, self.stride[0])
            return self._conv_forward(x, w, b)
        elif GEMM_3X3 and x.is_cuda and conv_gemm_func.eligible(x, self):
            if torch.is_autocast_enabled("cuda"):
                dt = torch.get_autocast_dtype("cuda")
                x, w, b = x.to(dt), w.to(dt), None if b is None else b.to(dt)
            if x.dtype == w.dtype and conv_gemm_func.eligible(x, self):
                return conv_gemm_func.conv3x3_gemm(x, w, b, self.stride[0], self.dilation[0])
        return self._conv_forward(x, w, b)


class Linear(nn.Linear):
    def forward(self, x):
        return linear(x, lookup(self.weight), None if self.bias is None else lookup(self.bias))


def cast_params_of(module):
    """fp32 parameters of every amp_cache.Conv2d / Linear below `module` (+ anything registered with `register`)."""
    out, seen = [], set()
    for m in module.modules():
        ps = []
        if isinstance(m, (Conv2d, Linear)):
            ps = [m.weight] + ([m.bias] if m.bias is not None else [])
        ps += list(getattr(m, "_amp_cache_extra", ()))
        for p in ps:
            if p.dtype == torch.float32 and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def register(module, *params):
    """Mark extra parameters of `module` (used through `lookup` in its forward) for the fused cast."""
    module._amp_cache_extra = tuple(params)


def set_groups(module, group_of):
    """Partition the fused cast of `module` into several FusedCast nodes: group_of(parameter name) -> int.  One node per group means
    one flat gradient buffer per group and -- what the segmented multi-GPU step of bench.py needs -- a backward pass restricted to
    some parameters no longer drags every layer that feeds the (single) cast node along.  None / never called: one group."""
    if group_of is None:
        module.__dict__.pop("_amp_cache_group_of", None)
    else:
        module.__dict__["_amp_cache_group_of"] = group_of
    module.__dict__.pop("_amp_cache_params", None)
    module.__dict__.pop("_amp_cache_plans", None)


def _param_groups(module):
    groups = module.__dict__.get("_amp_cache_params")
    if groups is None:
        params = cast_params_of(module)
        group_of = module.__dict__.get("_amp_cache_group_of")
        if group_of is None:
            groups = [params]
        else:
            names = {id(p): n for n, p in module.named_parameters()}
            by = {}
            for p in params:
                by.setdefault(int(group_of(names.get(id(p), ""))), []).append(p)
            groups = [by[k] for k in sorted(by)]
        module.__dict__["_amp_cache_params"] = groups
    return groups


@contextlib.contextmanager
def scope(module):
    """Inside: `lookup(p)` returns this forward's low-precision copy of p (when autocast is on for the GPU)."""
    groups = None
    if ENABLED and torch.is_autocast_enabled("cuda"):
        groups = [[p for p in g if p.is_cuda] for g in _param_groups(module)]
        groups = [g for g in groups if g]
    if not groups:
        yield
        return
    if _PARTIALS:       # a deferred partial sum of the LAST backward that did not come back through a one-launch cast lost its slices
        _PARTIALS.clear()
        raise RuntimeError("deferred weight-gradient partials were not consumed by the fused gradient cast")
    low = torch.get_autocast_dtype("cuda")
    for gi, params in enumerate(groups):
        plan = None
        if MULTI_CAST and low in (torch.bfloat16, torch.float16):
            plans = module.__dict__.setdefault("_amp_cache_plans", {})
            ptrs = [p.data_ptr() for p in params]
            plan = plans.get((low, gi))
            if plan is None or plan.src_ptrs_host != ptrs:      # first use, or parameters (re)allocated (.to(), memory format, ...)
                plan = plans[(low, gi)] = _CastPlan(params, low) if all(_dense(p) and p.dtype == torch.float32 for p in params) else _NoPlan(ptrs)
            if isinstance(plan, _NoPlan):
                plan = None
        outs = FusedCast.apply(low, plan, *params)
        if plan is not None:
            for p, o in zip(params, outs):
                if getattr(p, "_ocpg_single_use", False) and p.requires_grad:
                    o._ocpg_defer = True            # layers may leave partial sums of this gradient to FusedCast.backward
        _ACTIVE.update({id(p): o for p, o in zip(params, outs)})
    try:
        yield
    finally:
        _ACTIVE.clear()
