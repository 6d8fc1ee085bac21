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
