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


class WindowMeans3x3CL(Function):
    """WindowMeans3x3 for x [N,C,h,w] fp32 in CHANNELS-LAST memory (round 4), same result [N, C*9]."""

    @staticmethod
    def forward(ctx, x, as_bf16):
        xs = x.permute(0, 2, 3, 1)
        assert xs.is_contiguous() and xs.dtype == torch.float32
        n, h, w, c = xs.shape
        bands = int(lib().ocpg_window_sums3x3_cl_bands(h))
        part = torch.empty((n, bands, c * 9), dtype=torch.float32, device=x.device)
        check(lib().ocpg_window_sums3x3_cl(xs.data_ptr(), n, h, w, c, int(as_bf16), part.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_window_sums3x3_cl")
        ctx.shape = (n, c, h, w)
        return part.sum(1) * (1.0 / ((h - 2) * (w - 2)))

    @staticmethod
    @once_differentiable
    def backward(ctx, gm):
        n, c, h, w = ctx.shape
        gm = gm.float().contiguous()
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=gm.device)
        check(lib().ocpg_window_means3x3_bwd_cl(gm.data_ptr(), n, h, w, c, None, dx.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_window_means3x3_bwd_cl")
        return dx.permute(0, 3, 1, 2), None


def conv3x3_valid_spatial_mean(x, weight, bias, as_bf16):
    """== F.conv2d(x, weight, bias).mean(dim=(2, 3)) for a 3x3 valid convolution, without the convolution.  `weight` is the copy
    the convolution would have used (bf16 under autocast): the products are the same, only the summation order differs."""
    cl = x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] > 1 and x.permute(0, 2, 3, 1).is_contiguous() and x.shape[0] <= 65535
    m = WindowMeans3x3CL.apply(x, as_bf16) if cl else WindowMeans3x3.apply(x, as_bf16)
    with torch.autocast(device_type=x.device.type, enabled=False):
        return torch.nn.functional.linear(m, weight.float().flatten(1), None if bias is None else bias.float())


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class SpectralGateCL(Function):
    """The gate with its [Re || Im] side in CHANNELS-LAST memory and the 1x1 convs' compute dtype (csrc/spectral.hip c2p / p2c):
    the result is a [N, 2C, h, w] tensor whose memory is [N, h, w, 2C], so the following 1x1 convs are plain GEMMs on it and no
    cast / layout copy sits between the FFT and the GEMM."""

    @staticmethod
    def forward(ctx, spec, coef, high, dtype):
        spec = spec.contiguous()
        coef, high = coef.float().contiguous(), high.float().contiguous()
        n, c, h, w = spec.shape
        pair = torch.empty((n, h, w, 2 * c), dtype=dtype, device=spec.device)
        check(lib().ocpg_spectral_c2p(torch.view_as_real(spec).data_ptr(), coef.data_ptr(), high.data_ptr(), n, c, h * w, pair.data_ptr(),
                                      _DT[dtype], torch.cuda.current_stream().cuda_stream), "ocpg_spectral_c2p")
        ctx.save_for_backward(spec, coef, high)
        return pair.permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, gz):
        spec, coef, high = ctx.saved_tensors
        n, c, h, w = spec.shape
        g = gz.permute(0, 2, 3, 1).contiguous()
        if g.dtype not in _DT:
            g = g.float()
        dx = torch.empty((n, c, h, w, 2), dtype=torch.float32, device=spec.device)
        nb = ((c + 63) // 64) * ((h * w + 63) // 64)
        part = torch.empty((n, nb), dtype=torch.float32, device=spec.device)
        check(lib().ocpg_spectral_p2c(g.data_ptr(), torch.view_as_real(spec).data_ptr(), coef.data_ptr(), high.data_ptr(), n, c, h * w,
                                      dx.data_ptr(), part.data_ptr(), _DT[g.dtype], torch.cuda.current_stream().cuda_stream), "ocpg_spectral_p2c")
        return torch.view_as_complex(dx), part.sum(1), None, None


class PairToComplex(Function):
    """y [N, 2C, h, w] in channels-last memory (any of fp32 / bf16 / fp16) -> complex64 [N, C, h, w] = y[:, :C] + i y[:, C:]
    (`torch.complex(*torch.chunk(y.float(), 2, dim=1))`, models/modules.py:52-53) in one transposing pass each way."""

    @staticmethod
    def forward(ctx, y):
        n, c2, h, w = y.shape
        src = y.permute(0, 2, 3, 1).contiguous()
        if src.dtype not in _DT:
            src = src.float()
        out = torch.empty((n, c2 // 2, h, w, 2), dtype=torch.float32, device=y.device)
        check(lib().ocpg_spectral_p2c(src.data_ptr(), None, None, None, n, c2 // 2, h * w, out.data_ptr(), None, _DT[src.dtype],
                                      torch.cuda.current_stream().cuda_stream), "ocpg_spectral_p2c")
        ctx.dtype = y.dtype
        return torch.view_as_complex(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, gc):
        gc = gc.contiguous()
        n, c, h, w = gc.shape
        dt = ctx.dtype if ctx.dtype in _DT else torch.float32
        pair = torch.empty((n, h, w, 2 * c), dtype=dt, device=gc.device)
        check(lib().ocpg_spectral_c2p(torch.view_as_real(gc).data_ptr(), None, None, n, c, h * w, pair.data_ptr(), _DT[dt],
                                      torch.cuda.current_stream().cuda_stream), "ocpg_spectral_c2p")
        return pair.permute(0, 3, 1, 2).to(ctx.dtype)


def spectral_gate_cl(spec, coef, high, dtype):
    return SpectralGateCL.apply(spec, coef, high, dtype)


def pair_to_complex(y):
    return PairToComplex.apply(y)


class IFFT2Real(Function):
    """ifft2(z, s=(h, w)).real of a complex spectrum (models/modules.py:53-54) as one node.  autograd's backward of `.real` first builds
    a complex tensor (zeros + strided copy of the gradient: two passes over a [2560, 48, 80] complex map at the finest level) and runs
    a complex-to-complex transform on it; the gradient is REAL, so the same result is the real-to-complex transform
    fft2(g, norm="forward") -- half the butterflies and no complex staging."""

    @staticmethod
    def forward(ctx, z, h, w):
        ctx.sizes = (tuple(z.shape[-2:]), (h, w))
        return torch.fft.ifft2(z, s=(h, w)).real

    @staticmethod
    def backward(ctx, g):
        (zh, zw), (h, w) = ctx.sizes
        gz = torch.fft.fft2(g if g.dtype == torch.float32 else g.float(), norm="forward")
        if (zh, zw) != (h, w):          # ifft2 cropped / zero-padded its input to s: the mirror image on the way back
            gz = torch.nn.functional.pad(gz[..., :min(zh, h), :min(zw, w)], (0, max(zw - w, 0), 0, max(zh - h, 0)))
        return gz, None, None


def ifft2_real(z, h, w):
    return IFFT2Real.apply(z, int(h), int(w))


# ---- the transforms themselves on channels-last maps (csrc/lfm_dft.hip) ---------------------------------------------------------
def dft_supported(h, w):
    return bool(lib().ocpg_lfm_dft_supported(int(h), int(w)))


def _twiddles(n, device):
    """The tables of a line transform of length n (include/ocpg_hip.h): exp(-2 pi i m / n), then the L1- and the L2-point DFT matrices,
    formed in fp64: float2 [n + L1^2 + L2^2]"""
    from ....util.misc import memo

    def make():
        code = int(lib().ocpg_lfm_dft_split(int(n)))
        assert code, f"length {n} is not served by csrc/lfm_dft.hip"
        l1, l2 = code >> 8, code & 255

        def angles(idx, m):
            a = idx.to(torch.float64) * (-2.0 * torch.pi / m)
            return torch.stack([a.cos(), a.sin()], -1)
        parts = [angles(torch.arange(n), n)]
        for r in (l1, l2):
            k = torch.arange(r)
            parts.append(angles((k[:, None] * k[None, :]).remainder(r).flatten(), r))
        return torch.cat(parts).float().contiguous().to(device)
    return memo("lfm_tw", n, device, make)


def _nhwc32(x):
    """[N, C, h, w] -> its channels-last memory as an fp32 [N, h, w, C] contiguous tensor (a view when it already is)."""
    v = x.permute(0, 2, 3, 1)
    if v.dtype != torch.float32:
        v = v.float()
    return v if v.is_contiguous() else v.contiguous()


def _pair_cl(y):
    v = y.permute(0, 2, 3, 1)
    if v.dtype not in _DT:
        v = v.float()
    return v if v.is_contiguous() else v.contiguous()


def _spectrum(x_nhwc, coef, high, norm, dtype):
    n, h, w, c = x_nhwc.shape
    dev = x_nhwc.device
    tmp = torch.empty((n, h, w // 2 + 1, c, 2), dtype=torch.float32, device=dev)
    pair = torch.empty((n, h, w, 2 * c), dtype=dtype, device=dev)
    check(lib().ocpg_lfm_spectrum_fwd(x_nhwc.data_ptr(), coef.data_ptr() if coef is not None else None,
                                      high.data_ptr() if high is not None else None, n, h, w, c, _twiddles(h, dev).data_ptr(),
                                      _twiddles(w, dev).data_ptr(), float(norm), tmp.data_ptr(), pair.data_ptr(), _DT[dtype],
                                      torch.cuda.current_stream().cuda_stream), "ocpg_lfm_spectrum_fwd")
    return pair


def _inverse(pair, coef, high, z_saved, want_coef, norm, residual):
    n, h, w, c2 = pair.shape
    c = c2 // 2
    dev = pair.device
    tmp = torch.empty((n, h, w // 2 + 1, c, 2), dtype=torch.float32, device=dev)
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)
    part = torch.empty((n, (w // 2 + 1) * ((c + 63) // 64)), dtype=torch.float32, device=dev) if want_coef else None
    check(lib().ocpg_lfm_spectrum_inv(pair.data_ptr(), _DT[pair.dtype], coef.data_ptr() if coef is not None else None,
                                      high.data_ptr() if high is not None else None, z_saved.data_ptr() if want_coef else None,
                                      part.data_ptr() if want_coef else None, n, h, w, c, _twiddles(h, dev).data_ptr(), _twiddles(w, dev).data_ptr(),
                                      float(norm), tmp.data_ptr(), residual.data_ptr() if residual is not None else None, out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "ocpg_lfm_spectrum_inv")
    return out, part


class LFMSpectrum(Function):
    """z = cat([Re, Im], 1)(fft2(x) * (1 - coef * high)) of a real map (models/modules.py:44-50) as a [N, 2C, h, w] tensor in
    channels-last memory and the 1x1 convs' compute dtype: two passes of csrc/lfm_dft.hip.  Backward: the real part of the unnormalised
    inverse transform of gate * gz, and the gate coefficient's gradient from the saved z."""

    @staticmethod
    def forward(ctx, x, coef, high, dtype):
        xs = _nhwc32(x)
        coef, high = coef.float().contiguous(), high.float().contiguous()
        pair = _spectrum(xs, coef, high, 1.0, dtype)
        ctx.save_for_backward(pair, coef, high)
        return pair.permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, gz):
        pair, coef, high = ctx.saved_tensors
        g = _pair_cl(gz)
        if g.dtype != pair.dtype:
            g = g.to(pair.dtype)
        want_coef = ctx.needs_input_grad[1]
        gx, part = _inverse(g, coef, high, pair, want_coef, 1.0, None)
        return gx.permute(0, 3, 1, 2), (part.sum(1) if want_coef else None), None, None


class LFMInverse(Function):
    """x + ifft2(complex(y[:, :C], y[:, C:]), s=(h, w)).real (models/modules.py:52-56) for y in channels-last memory, as two passes of
    csrc/lfm_dft.hip; fp32 result in channels-last memory.  Backward: fft2(g) / (h w) as a [Re || Im] pair in y's dtype; g for x."""

    @staticmethod
    def forward(ctx, y, x):
        ys, xs = _pair_cl(y), _nhwc32(x)
        n, h, w, _ = ys.shape
        out, _ = _inverse(ys, None, None, None, False, 1.0 / (h * w), xs)
        ctx.dtype = (y.dtype, ys.dtype)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ydt, sdt = ctx.dtype
        gy = None
        if ctx.needs_input_grad[0]:
            gs = _nhwc32(g)
            n, h, w, _ = gs.shape
            gy = _spectrum(gs, None, None, 1.0 / (h * w), sdt).permute(0, 3, 1, 2)
            if gy.dtype != ydt:
                gy = gy.to(ydt)
        return gy, (g if ctx.needs_input_grad[1] else None)


def lfm_spectrum(x, coef, high, dtype):
    return LFMSpectrum.apply(x, coef, high, dtype)


def lfm_inverse(y, x):
    return LFMInverse.apply(y, x)
