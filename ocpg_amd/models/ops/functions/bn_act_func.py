"""Fused frozen-BN affine (+ residual) (+ ReLU): autograd binding of ocpg_bn_act_{fwd,bwd} (csrc/bn_act.hip).

One HBM pass replaces FrozenBatchNorm2d.forward's elementwise chain (models/backbone.py:46-56) plus the Bottleneck's
residual add and ReLU.  Only the OUTPUT is saved for backward (it is the next conv's input anyway).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib, stream_ptr

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _layout(x):
    """-> (n_outer, C, inner) of a 4-D map in its actual memory layout, or None if it is neither NHWC nor NCHW dense."""
    n, c, h, w = x.shape
    if x.is_contiguous(memory_format=torch.channels_last) and not (c == 1 or h * w == 1):
        return n * h * w, c, 1
    if x.is_contiguous():
        return n, c, h * w
    if x.is_contiguous(memory_format=torch.channels_last):
        return n * h * w, c, 1
    return None


class FrozenBNAct(Function):
    @staticmethod
    def forward(ctx, x, scale, shift, skip, relu):
        if not x.is_cuda:
            raise RuntimeError("FrozenBNAct: x must be a GPU tensor: Not implemented on the CPU")
        if x.dtype not in _DT:
            raise RuntimeError(f"FrozenBNAct: unsupported dtype {x.dtype}")
        lay = _layout(x)
        if lay is None:
            x = x.contiguous(memory_format=torch.channels_last)
            lay = _layout(x)
        if skip is not None:
            if skip.dtype != x.dtype:
                skip = skip.to(x.dtype)
            if _layout(skip) != lay or skip.stride() != x.stride():
                skip = skip.contiguous(memory_format=torch.channels_last if lay[2] == 1 else torch.contiguous_format)
                if skip.stride() != x.stride():
                    x = x.contiguous(memory_format=torch.channels_last if lay[2] == 1 else torch.contiguous_format)
        y = torch.empty_like(x)          # preserves the memory format
        assert y.stride() == x.stride()
        n_outer, C, inner = lay
        with torch.cuda.device(x.device):
            check(lib().ocpg_bn_act_fwd(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), skip.data_ptr() if skip is not None else None,
                                        y.data_ptr(), n_outer, C, inner, int(relu), _DT[x.dtype], stream_ptr()), "ocpg_bn_act_fwd")
        ctx.save_for_backward(y, scale)
        ctx.meta = (lay, bool(relu), skip is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        lay, relu, has_skip = ctx.meta
        need_x, need_skip = ctx.needs_input_grad[0], has_skip and ctx.needs_input_grad[3]
        if gy.stride() != y.stride() or gy.dtype != y.dtype:
            gy = gy.to(y.dtype).contiguous(memory_format=torch.channels_last if lay[2] == 1 else torch.contiguous_format)
            if gy.stride() != y.stride():
                gy = gy.as_strided(y.shape, y.stride()).clone() if gy.numel() == 0 else torch.empty_like(y).copy_(gy)
        gx = torch.empty_like(y) if need_x else None
        gskip = torch.empty_like(y) if need_skip else None
        if need_x or need_skip:
            n_outer, C, inner = lay
            with torch.cuda.device(y.device):
                check(lib().ocpg_bn_act_bwd(gy.data_ptr(), y.data_ptr(), scale.data_ptr(), gx.data_ptr() if gx is not None else None,
                                            gskip.data_ptr() if gskip is not None else None, n_outer, C, inner, int(relu),
                                            _DT[y.dtype], stream_ptr()), "ocpg_bn_act_bwd")
        return gx, None, None, gskip, None


def frozen_bn_act(x, scale, shift, skip=None, relu=True):
    return FrozenBNAct.apply(x, scale, shift, skip, relu)
