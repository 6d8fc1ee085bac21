"""Autograd bindings of csrc/fused_ln.hip: LayerNorm(res + dropout(x)) and dropout(relu(x W^T + b)) as single passes.

The dropout masks are never stored: forward and backward evaluate the same counter-based generator on (seed, offset);
seed = torch.initial_seed() (so torch.manual_seed governs it), offset = a per-process call counter.  The counter is
state OUTSIDE torch's generators: util/checkpoint.py saves / restores it (get_rng_state / set_rng_state below), so a resumed
run continues the mask sequence instead of replaying it from offset 1.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib
from .gemm_func import mm

_DT = {torch.float32: 0, torch.bfloat16: 1}
_calls = [0]


def _rng():
    _calls[0] += 1
    return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, _calls[0]


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
        seed, offset = rng if rng is not None else _rng()
        y = torch.empty((r, c), dtype=torch.float32, device=x.device)
        stats = torch.empty((2, r), dtype=torch.float32, device=x.device)
        check(lib().ocpg_dropout_add_ln_fwd(x2.data_ptr(), res2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), r, c, float(eps), float(p), seed,
                                            offset, _DT[x2.dtype], y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), _st()),
              "ocpg_dropout_add_ln_fwd")
        ctx.save_for_backward(x2, res2, gamma, stats)
        ctx.meta = (float(p), seed, offset, x.shape, res.shape)
        return y.view(res.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, res2, gamma, stats = ctx.saved_tensors
        p, seed, offset, xshape, rshape = ctx.meta
        r, c = x2.shape
        gy = gy.reshape(r, c).float().contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gres = torch.empty_like(res2) if ctx.needs_input_grad[1] else None
        part = torch.empty((lib().ocpg_dropout_add_ln_bwd_slots(r), 2, c), dtype=torch.float32, device=x2.device)
        check(lib().ocpg_dropout_add_ln_bwd(gy.data_ptr(), x2.data_ptr(), res2.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                            r, c, p, seed, offset, _DT[x2.dtype], None if gx is None else gx.data_ptr(),
                                            None if gres is None else gres.data_ptr(), part.data_ptr(), _st()),
              "ocpg_dropout_add_ln_bwd")
        dgb = part.sum(0)
        return (None if gx is None else gx.view(xshape), None if gres is None else gres.view(rshape), dgb[0], dgb[1], None, None, None)


class LinearBiasReluDropout(Function):
    @staticmethod
    def forward(ctx, x2, w, b, p, rng, splits):
        seed, offset = rng if rng is not None else _rng()
        h = mm(x2, w, True)
        r, c = h.shape
        check(lib().ocpg_bias_relu_dropout_fwd(h.data_ptr(), b.data_ptr(), r, c, float(p), seed, offset, _DT[h.dtype], h.data_ptr(), _st()),
              "ocpg_bias_relu_dropout_fwd")
        ctx.save_for_backward(x2, w, h)
        ctx.meta = (float(p), splits)
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
        dbias = part.sum(0)
        gx = mm(ga, w) if ctx.needs_input_grad[0] else None
        gw = weight_grad(ga, x2) if ctx.needs_input_grad[1] else None
        return gx, gw, dbias.to(h.dtype) if ctx.needs_input_grad[2] else None, None, None, None


def supported(x, res, c):
    return (x.is_cuda and x.dtype in _DT and res.dtype == torch.float32 and c % 4 == 0 and c <= 2048 and x.shape == res.shape)


def dropout_add_layer_norm(x, res, norm, p, rng=None):
    """norm(res + dropout_p(x)) for an nn.LayerNorm over the last axis; res fp32, x fp32/bf16; -> fp32."""
    return DropoutAddLayerNorm.apply(x, res, norm.weight, norm.bias, p, norm.eps, rng)
