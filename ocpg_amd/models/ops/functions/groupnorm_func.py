"""Autograd binding of csrc/groupnorm.hip: the GroupNorm behind the input projections (reference models/ocpg.py:108-119), reading the
channels-last map the projection GEMM wrote and writing the fp32 planes the LFM's FFTs read -- one launch each way."""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib

_CODE = {torch.float32: 1, torch.bfloat16: 0, torch.float16: 2}


def _scratch(x, groups):
    """Per-tile partials of the tiled kernels (large maps); None when the one-launch kernels are used."""
    n, c, h, w = x.shape
    words = lib().ocpg_groupnorm_cl_work(n, h * w, c, groups)
    return torch.empty(words, dtype=torch.float32, device=x.device) if words > 0 else None


CL_OUT = os.environ.get("OCPG_GN_CL_OUT", "1") != "0"     # A/B switch: the fp32 output channels-last as well (the LFM's own transforms read that; "0": planes, as rocFFT wanted)


class GroupNormCLFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, cl_out=False):
        n, c, h, w = x.shape
        y = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2) if cl_out else \
            torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        mean = torch.empty((n, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty((n, groups), dtype=torch.float32, device=x.device)
        work = _scratch(x, groups)
        fwd = lib().ocpg_groupnorm_cl2cl_fwd if cl_out else lib().ocpg_groupnorm_cl_fwd
        check(fwd(x.data_ptr(), _CODE[x.dtype], weight.data_ptr(), bias.data_ptr(), n, h * w, c, groups, float(eps),
                                          y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0 if work is None else work.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "ocpg_groupnorm_cl_fwd")
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.groups, ctx.cl_out = groups, cl_out
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, mean, rstd = ctx.saved_tensors
        n, c, h, w = x.shape
        if ctx.cl_out:
            gy = gy.float().permute(0, 2, 3, 1)
            gy = gy if gy.is_contiguous() else gy.contiguous()
        else:
            gy = gy.float().contiguous()
        dx = torch.empty_like(x)                       # preserves the channels-last strides
        part = torch.empty((n, 2, c), dtype=torch.float32, device=x.device)
        work = _scratch(x, ctx.groups)
        bwd = lib().ocpg_groupnorm_cl2cl_bwd if ctx.cl_out else lib().ocpg_groupnorm_cl_bwd
        check(bwd(gy.data_ptr(), x.data_ptr(), _CODE[x.dtype], weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), n,
                                          h * w, c, ctx.groups, dx.data_ptr(), part.data_ptr(), 0 if work is None else work.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "ocpg_groupnorm_cl_bwd")
        dgb = part.sum(0)                              # one reduction for dgamma and dbeta
        return dx, dgb[0], dgb[1], None, None, None


def eligible(x, module):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _CODE and module.affine and module.num_channels == 8 * module.num_groups
            and module.weight.dtype == torch.float32 and x.shape[0] > 0 and x.shape[2] * x.shape[3] > 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0)      # 16-byte pixel-octet loads


class GroupNorm(torch.nn.GroupNorm):
    """nn.GroupNorm (same parameters / state_dict).  For a channels-last GPU map with groups of 8 channels (the reference's
    GroupNorm(32, 256) on a projection output) the HIP pass; any other input takes ATen's group_norm.  The output is fp32
    either way (autocast runs group_norm in fp32): channels-last when the LFM block behind it transforms this map size with its own
    passes (csrc/lfm_dft.hip reads channels-last), planes when it keeps rocFFT (lengths with a prime factor > 16, e.g. config #5's
    107-wide level: a channels-last map would cost that path two transposing copies)."""

    def forward(self, x):
        if eligible(x, self):
            cl = CL_OUT and bool(lib().ocpg_lfm_dft_supported(int(x.shape[2]), int(x.shape[3])))
            return GroupNormCLFunction.apply(x, self.weight, self.bias, self.num_groups, self.eps, cl)
        return super().forward(x)
