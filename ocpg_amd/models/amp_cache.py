"""One fused cast of all autocast-consumed parameters per forward.

Under autocast every Conv2d / Linear casts its fp32 weight (and bias) to bf16/fp16 with its own kernel and its own
autograd node, and the backward casts every weight gradient back with another kernel: ~200 + ~200 launches per step for
ResNet-101 + the transformer -- on MI355X the training step is host/launch-bound, so they matter.
Here the model casts ALL such parameters once per forward with a multi-tensor copy (`FusedCast`, one autograd node whose
backward is one multi-tensor copy of the gradients back to fp32), and the layers below pick the low-precision copies up
through `lookup`.  Outside autocast nothing changes (the table is empty and `lookup` returns the parameter itself).
Numerically identical to autocast's own per-op casts.
"""
import contextlib

import torch
import torch.nn.functional as F
from torch import nn

_ACTIVE = {}        # id(fp32 parameter) -> low-precision copy, valid inside one `scope`
ENABLED = True      # A/B switch (tests compare against autocast's own per-op casts)


class FusedCast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dtype, *params):
        outs = [torch.empty_like(p, dtype=dtype) for p in params]
        torch._foreach_copy_(outs, list(params))
        ctx.src_dtype = params[0].dtype
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        idx = [i for i, g in enumerate(grads) if g is not None and ctx.needs_input_grad[i + 1]]
        res = [None] * len(grads)
        if idx:
            outs = [torch.empty_like(grads[i], dtype=ctx.src_dtype) for i in idx]
            torch._foreach_copy_(outs, [grads[i] for i in idx])
            for i, o in zip(idx, outs):
                res[i] = o
        return (None, *res)


def lookup(p):
    """The low-precision copy of parameter `p` for the current forward, or `p` itself."""
    return _ACTIVE.get(id(p), p) if _ACTIVE else p


class Conv2d(nn.Conv2d):
    def forward(self, x):
        return self._conv_forward(x, lookup(self.weight), None if self.bias is None else lookup(self.bias))


class Linear(nn.Linear):
    def forward(self, x):
        return F.linear(x, lookup(self.weight), None if self.bias is None else lookup(self.bias))


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


@contextlib.contextmanager
def scope(module):
    """Inside: `lookup(p)` returns this forward's low-precision copy of p (when autocast is on for the GPU)."""
    params = None
    if ENABLED and torch.is_autocast_enabled("cuda"):
        params = module.__dict__.get("_amp_cache_params")
        if params is None:
            params = module.__dict__["_amp_cache_params"] = cast_params_of(module)
        params = [p for p in params if p.is_cuda]
    if not params:
        yield
        return
    outs = FusedCast.apply(torch.get_autocast_dtype("cuda"), *params)
    _ACTIVE.update({id(p): o for p, o in zip(params, outs)})
    try:
        yield
    finally:
        _ACTIVE.clear()
