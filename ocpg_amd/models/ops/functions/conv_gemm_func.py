"""3x3 convolution of a channels-last map as im2col (HIP, csrc/im2col.hip) + one hipBLASLt GEMM, with its backward.

Replaces the MIOpen kernels behind nn.Conv2d(k=3, padding=dilation, stride 1|2) where the input is channels-last on the
GPU (torchvision Bottleneck.conv2 in the reference's backbone, models/backbone.py:86-117; neck convs ocpg.py:118-126).
Backward: dcols = gy W (GEMM) -> col2im (HIP) for the input gradient; weight gradient = gy^T cols as a row-split
batched GEMM (amp_cache.weight_grad) on the SAVED patch matrix (288 GB of HBM: ~2 GB for all of ResNet-101 at config #2).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib, stream_ptr
from .gemm_func import mm

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
ALWAYS = False      # tests: take the GEMM path for every eligible geometry, not only where it is faster


def eligible(x, conv):
    """3x3, padding == dilation, stride 1|2, groups 1, channels-last GPU input whose pixels are 16-byte multiples."""
    g = conv.__dict__.get("_gemm3x3")
    if g is None:
        g = conv.__dict__["_gemm3x3"] = (conv.kernel_size == (3, 3) and conv.groups == 1 and conv.padding_mode == "zeros"
                                         and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
                                         and conv.dilation[0] == conv.dilation[1] and conv.padding == conv.dilation)
    if not (g and x.is_cuda and x.dim() == 4 and x.dtype in _DT and (x.shape[1] * x.element_size()) % 16 == 0
            and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    if ALWAYS:
        return True
    # where it pays (measured, tools/bench_conv3x3.py, fwd+bwd GPU-busy, bf16, 10 frames): 256ch 24x40 116 vs 133 us,
    # 512ch 24x40/s2 105 vs 151, 512ch 12x20 102 vs 151; it loses on the large maps (64ch 96x160: 265 vs 131 us), where
    # the 9x patch matrix makes the GEMM memory-bound
    s = conv.stride[0]
    rows = x.shape[0] * ((x.shape[2] - 1) // s + 1) * ((x.shape[3] - 1) // s + 1)
    return x.shape[1] >= 256 and rows <= 12288


class Conv3x3AsGemm(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, dil):
        n, c, h, wd = x.shape
        co = w.shape[0]
        ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
        cols = torch.empty((n * ho * wo, 9 * c), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            check(lib().ocpg_im2col3x3_nhwc(x.data_ptr(), n, h, wd, c, stride, dil, cols.data_ptr(), _DT[x.dtype], stream_ptr()),
                  "ocpg_im2col3x3_nhwc")
        w2 = w.permute(0, 2, 3, 1).reshape(co, 9 * c)             # a view when the weight is channels-last
        y2 = mm(cols, w2, True, bias)
        ctx.save_for_backward(cols, w)
        ctx.geom = (n, c, h, wd, ho, wo, stride, dil, bias is not None)
        return y2.view(n, ho, wo, co).permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        from ...amp_cache import weight_grad
        cols, w = ctx.saved_tensors
        n, c, h, wd, ho, wo, stride, dil, has_bias = ctx.geom
        co = w.shape[0]
        gy2 = gy.permute(0, 2, 3, 1).reshape(n * ho * wo, co)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            dcols = mm(gy2, w.permute(0, 2, 3, 1).reshape(co, 9 * c))
            gx_nhwc = torch.empty((n, h, wd, c), dtype=gy.dtype, device=gy.device)
            with torch.cuda.device(gy.device):
                check(lib().ocpg_col2im3x3_nhwc(dcols.data_ptr(), n, h, wd, c, stride, dil, gx_nhwc.data_ptr(), _DT[gy.dtype],
                                                stream_ptr()), "ocpg_col2im3x3_nhwc")
            gx = gx_nhwc.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = weight_grad(gy2, cols).view(co, 3, 3, c).permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy2.sum(0)
        return gx, gw, gb, None, None


def conv3x3_gemm(x, w, bias, stride, dil):
    return Conv3x3AsGemm.apply(x, w, bias, int(stride), int(dil))
