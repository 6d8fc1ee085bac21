"""Autograd binding of csrc/spectral.hip: the LFM block's spectral gate as one pass each way."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib


class SpectralGate(Function):
    @staticmethod
    def forward(ctx, spec, coef, high):
        """spec complex64 [N,C,h,w], coef [N], high [h,w] (no gradient) -> fp32 [N,2C,h,w] = cat(Re, Im)(spec * (1 - coef*high))."""
        spec = spec.contiguous()
        coef, high = coef.float().contiguous(), high.float().contiguous()
        n, c, h, w = spec.shape
        out = torch.empty((n, 2 * c, h, w), dtype=torch.float32, device=spec.device)
        check(lib().ocpg_spectral_gate_fwd(torch.view_as_real(spec).data_ptr(), coef.data_ptr(), high.data_ptr(), n, c, h * w, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "ocpg_spectral_gate_fwd")
        ctx.save_for_backward(spec, coef, high)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        spec, coef, high = ctx.saved_tensors
        n, c, h, w = spec.shape
        gout = gout.float().contiguous()
        dx = torch.empty((n, c, h, w, 2), dtype=torch.float32, device=spec.device)
        part = torch.empty((n, c * ((h * w + 255) // 256)), dtype=torch.float32, device=spec.device)
        check(lib().ocpg_spectral_gate_bwd(gout.data_ptr(), torch.view_as_real(spec).data_ptr(), coef.data_ptr(), high.data_ptr(), n, c, h * w,
                                           dx.data_ptr(), part.data_ptr(), torch.cuda.current_stream().cuda_stream), "ocpg_spectral_gate_bwd")
        return torch.view_as_complex(dx), part.sum(1), None


def spectral_gate(spec, coef, high):
    return SpectralGate.apply(spec, coef, high)


class WindowMeans3x3(Function):
    """x [N,C,h,w] fp32 -> [N, C*9]: mean of every plane over the nine (h-2)x(w-2) windows at offsets (ky,kx) (csrc/lfm.hip)."""

    @staticmethod
    def forward(ctx, x, as_bf16):
        x = x.contiguous()
        n, c, h, w = x.shape
        out = torch.empty((n, c * 9), dtype=torch.float32, device=x.device)
        check(lib().ocpg_window_means3x3_fwd(x.data_ptr(), n * c, h, w, int(as_bf16), out.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_window_means3x3_fwd")
        ctx.shape = (n, c, h, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gm):
        n, c, h, w = ctx.shape
        gm = gm.float().contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=gm.device)
        check(lib().ocpg_window_means3x3_bwd(gm.data_ptr(), n * c, h, w, dx.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_window_means3x3_bwd")
        return dx, None


def conv3x3_valid_spatial_mean(x, weight, bias, as_bf16):
    """== F.conv2d(x, weight, bias).mean(dim=(2, 3)) for a 3x3 valid convolution, without the convolution.  `weight` is the copy
    the convolution would have used (bf16 under autocast): the products are the same, only the summation order differs."""
    m = WindowMeans3x3.apply(x, as_bf16)
    with torch.autocast(device_type=x.device.type, enabled=False):
        return torch.nn.functional.linear(m, weight.float().flatten(1), None if bias is None else bias.float())
