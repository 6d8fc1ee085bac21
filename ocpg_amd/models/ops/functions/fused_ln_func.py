"""Autograd bindings of csrc/fused_ln.hip: LayerNorm(res + dropout(x)) and dropout(relu(x W^T + b)) as single passes.

The dropout masks are never stored: forward and backward evaluate the same counter-based generator on (seed, offset);
seed = torch.initial_seed() (so torch.manual_seed governs it), offset = a per-process call counter.  The counter is
state OUTSIDE torch's generators: util/checkpoint.py saves / restores it (get_rng_state / set_rng_state below), so a resumed
run continues the mask sequence instead of replaying it from offset 1.

HIP graphs: a captured launch bakes its by-value arguments into the node, so a replayed step would draw the SAME masks every
time.  Calls made while a `GraphRng` is active therefore also hand the kernels a DEVICE pointer to a per-capture base word that
the kernel adds to its offset; the captured step ends with `advance()` (base += calls per replay, one tiny captured kernel), so
replay r of a graph captured at host counter c0 uses offsets c0 + r * J + j -- exactly the offsets eager step r would have used
(tests/test_graph_gpu.py: consecutive replays differ, replay k == eager step k).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib
from .gemm_func import mm

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_calls = [0]
_active = [None]            # the GraphRng whose capture is in progress (at most one per process)


class GraphRng:
    """Per-capture device state of the dropout generator: state[0] = base added to every captured call's offset when the
    kernel runs, state[1] = calls per replay.  Use:  `with rng: <capture, ending with rng.advance()>`, then `rng.finalize()`
    once after the capture and `rng.replayed()` after every replay (keeps the host counter, which checkpoints save, in step)."""

    def __init__(self, device):
        self.state = torch.zeros(2, dtype=torch.int64, device=device)
        self.c0, self.calls = None, 0

    def __enter__(self):
        assert _active[0] is None, "nested graph captures of the dropout generator"
        self.c0 = _calls[0]
        _active[0] = self
        return self

    def __exit__(self, *exc):
        self.calls = _calls[0] - self.c0
        _calls[0] = self.c0         # a capture executes nothing: the first replay is the step that uses offsets c0 + 1 ...
        _active[0] = None
        return False

    def advance(self):
        """LAST operation inside the capture: base += calls per replay (state[1] is filled by finalize())."""
        self.state[0:1].add_(self.state[1:2])

    def finalize(self):
        self.state[1] = self.calls

    def replayed(self):
        _calls[0] += self.calls


def _rng():
    """(seed, offset, base pointer or None) of the next dropout call."""
    _calls[0] += 1
    g = _active[0]
    return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, _calls[0], (None if g is None else g.state.data_ptr())


def _unpack(rng):
    """Explicit generators (tests): (seed, offset) or (seed, offset, base pointer)."""
    rng = rng if rng is not None else _rng()
    return (rng[0], rng[1], rng[2] if len(rng) > 2 else None)


def get_rng_state():
    return {"dropout_calls": _calls[0]}


def set_rng_state(state):
    _calls[0] = int(state.get("dropout_calls", 0))


def _st():
    return torch.cuda.current_stream().cuda_stream


class DropoutAddLayerNorm(Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, p, eps, rng):
        c = x.shape[-1]
        x2, res2 = x.reshape(-1, c).contiguous(), res.reshape(-1, c).contiguous()
        r = x2.shape[0]
        seed, offset, base = _unpack(rng)
        y = torch.empty((r, c), dtype=torch.float32, device=x.device)
        stats = torch.empty((2, r), dtype=torch.float32, device=x.device)
        check(lib().ocpg_dropout_add_ln_fwd(x2.data_ptr(), res2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), r, c, float(eps), float(p), seed,
                                            offset, base, _DT[x2.dtype], y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), _st()),
              "ocpg_dropout_add_ln_fwd")
        ctx.save_for_backward(x2, res2, gamma, stats)
        ctx.meta = (float(p), seed, offset, base, x.shape, res.shape)
        return y.view(res.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, res2, gamma, stats = ctx.saved_tensors
        p, seed, offset, base, xshape, rshape = ctx.meta
        r, c = x2.shape
        gy = gy.reshape(r, c).float().contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gres = torch.empty_like(res2) if ctx.needs_input_grad[1] else None
        part = torch.empty((lib().ocpg_dropout_add_ln_bwd_slots(r), 2, c), dtype=torch.float32, device=x2.device)
        check(lib().ocpg_dropout_add_ln_bwd(gy.data_ptr(), x2.data_ptr(), res2.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                            r, c, p, seed, offset, base, _DT[x2.dtype], None if gx is None else gx.data_ptr(),
                                            None if gres is None else gres.data_ptr(), part.data_ptr(), _st()),
              "ocpg_dropout_add_ln_bwd")
        dgb = part.sum(0)
        return (None if gx is None else gx.view(xshape), None if gres is None else gres.view(rshape), dgb[0], dgb[1], None, None, None)


class LinearBiasReluDropout(Function):
    @staticmethod
    def forward(ctx, x2, w, b, p, rng, splits):
        seed, offset, base = _unpack(rng)
        h = mm(x2, w, True)
        r, c = h.shape
        check(lib().ocpg_bias_relu_dropout_fwd(h.data_ptr(), b.data_ptr(), r, c, float(p), seed, offset, base, _DT[h.dtype], h.data_ptr(), _st()),
              "ocpg_bias_relu_dropout_fwd")
        ctx.save_for_backward(x2, w, h)
        ctx.meta = (float(p), splits)
        from ...amp_cache import deferrable
        ctx.defer_w, ctx.defer_b = deferrable(w), deferrable(b) and b.dtype == h.dtype
        return h

    @staticmethod
    @once_differentiable
    def backward(ctx, gh):
        from ...amp_cache import weight_grad
        x2, w, h = ctx.saved_tensors
        p, splits = ctx.meta
        r, c = h.shape
        gh = gh.to(h.dtype).contiguous()
        ga = torch.empty_like(h)
        slots = lib().ocpg_bias_relu_dropout_bwd_slots(r, c, _DT[h.dtype])
        part = torch.empty((slots, c), dtype=torch.float32, device=h.device)
        check(lib().ocpg_bias_relu_dropout_bwd(gh.data_ptr(), h.data_ptr(), r, c, p, _DT[h.dtype], ga.data_ptr(), part.data_ptr(), _st()),
              "ocpg_bias_relu_dropout_bwd")
        from ...amp_cache import defer_sum
        # the per-slot partial column sums: summed inside the fused gradient cast when the bias copy allows it (one launch less + no cast)
        dbias = defer_sum(part, h.dtype) if (ctx.defer_b and ctx.needs_input_grad[2] and h.dtype != torch.float32) else part.sum(0).to(h.dtype)
        gx = mm(ga, w) if ctx.needs_input_grad[0] else None
        gw = weight_grad(ga, x2, ctx.defer_w) if ctx.needs_input_grad[1] else None
        return gx, gw, dbias if ctx.needs_input_grad[2] else None, None, None, None


def supported(x, res, c):
    return (x.is_cuda and x.dtype in _DT and res.dtype == torch.float32 and c % 4 == 0 and c <= 2048 and x.shape == res.shape)


def dropout_add_layer_norm(x, res, norm, p, rng=None):
    """norm(res + dropout_p(x)) for an nn.LayerNorm over the last axis; res fp32, x fp32/bf16; -> fp32."""
    return DropoutAddLayerNorm.apply(x, res, norm.weight, norm.bias, p, norm.eps, rng)
