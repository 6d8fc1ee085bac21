"""MSO's convolutions and its x2 resize on channels-last maps (csrc/mso.hip) as autograd nodes.

conv3x3_n16(x, w, ...) = conv2d(relu?(x), w, padding=1) (+ bias) (+ addend broadcast over groups of images) (+ residual) with
x [NB, H, W, C], w [co <= 16, 9, C] (tap-major: the reference's [co, C, 3, 3] weight permuted to (0, 2, 3, 1)), fp32 output
[NB, H, W, co].  Reference: models/decoder.py:22-46 (`conv1_1div8/4`, `conv2_1div8/4`, `out_conv`, F.interpolate).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_TORCH = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


def compute_code(device_type="cuda"):
    """The convolution's operand type: the autocast dtype when autocast is on (fp32 accumulation either way), else fp32."""
    if torch.is_autocast_enabled(device_type):
        return _DT.get(torch.get_autocast_dtype(device_type), 0)
    return 0


def _ptr(t):
    return None if t is None else t.data_ptr()


def _conv(x, relu_in, w, bias, addend, mask, residual, out_dtype, cdt):
    nb, h, wd, c = x.shape
    co = w.shape[0]
    out = torch.empty((nb, h, wd, co), dtype=out_dtype, device=x.device)
    rc = lib().ocpg_mso_conv3x3(x.data_ptr(), _DT[x.dtype], int(relu_in), w.data_ptr(), _DT[w.dtype], _ptr(bias), _ptr(addend),
                                1 if addend is None else addend.shape[0], _ptr(mask), 0 if mask is None else _DT[mask.dtype], _ptr(residual),
                                out.data_ptr(), _DT[out_dtype], nb, h, wd, c, co, cdt, torch.cuda.current_stream().cuda_stream)
    if rc:
        check(rc, "ocpg_mso_conv3x3")
    return out


class Conv3x3N16(Function):
    @staticmethod
    def forward(ctx, x, w, bias, addend, residual, relu_in, cdt):
        assert x.dim() == 4 and x.is_contiguous() and x.dtype in _DT and w.dim() == 3 and w.shape[1] == 9 and w.shape[2] == x.shape[3]
        assert w.shape[0] <= 16 and w.is_contiguous() and w.dtype in _DT
        co = w.shape[0]
        for t, shape in ((bias, (co,)), (residual, (*x.shape[:3], co))):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape)
        if addend is not None:
            assert addend.dtype == torch.float32 and addend.is_contiguous() and tuple(addend.shape[1:]) == (*x.shape[1:3], co)
            assert x.shape[0] % addend.shape[0] == 0
        out = _conv(x, relu_in, w, bias, addend, None, residual, torch.float32, cdt)
        ctx.save_for_backward(x, w)
        ctx.meta = (bool(relu_in), cdt, bias is not None, None if addend is None else addend.shape[0], residual is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        relu_in, cdt, has_bias, na, has_res = ctx.meta
        nb, h, wd, c = x.shape
        co = w.shape[0]
        if go.dtype != torch.float32 or not go.is_contiguous():
            go = go.float().contiguous()
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        gx = gw = gb = ga = None
        if ctx.needs_input_grad[0]:
            # the same convolution with flipped taps and the channel axes swapped; ReLU'(x) as the output mask
            wt = w.view(co, 3, 3, c).flip(1, 2).permute(3, 1, 2, 0).reshape(c, 9, co).to(_TORCH[cdt])      # .to / reshape: one contiguous copy
            if not wt.is_contiguous():
                wt = wt.contiguous()
            gx = _conv(go, False, wt, None, None, x if relu_in else None, None, x.dtype, cdt)
        if ctx.needs_input_grad[1]:
            rows = int(L.ocpg_mso_wgrad_rows(nb, h, c, cdt))
            bands = nb * ((h + rows - 1) // rows)
            part = torch.empty((bands, co, 9, c), dtype=torch.float32, device=x.device)
            want_b = has_bias and ctx.needs_input_grad[2]
            part_b = torch.empty((bands, 16), dtype=torch.float32, device=x.device) if want_b else None
            rc = L.ocpg_mso_wgrad(x.data_ptr(), _DT[x.dtype], int(relu_in), go.data_ptr(), part.data_ptr(), _ptr(part_b), nb, h, wd, c, co, rows,
                                  cdt, st)
            if rc:
                check(rc, "ocpg_mso_wgrad")
            gw = part.sum(0).to(w.dtype)
            if want_b:          # the band sums of g rode along in the weight-gradient kernel
                gb = part_b.sum(0)[:co]
        if gb is None and has_bias and ctx.needs_input_grad[2]:
            gb = go.view(-1, co).sum(0)
        if na is not None and ctx.needs_input_grad[3]:
            ga = go if na == nb else go.view(nb // na, na, h, wd, co).sum(0)
        return gx, gw, gb, ga, (go if has_res and ctx.needs_input_grad[4] else None), None, None


def conv3x3_n16(x, w, bias=None, addend=None, residual=None, relu_in=False, cdt=0):
    return Conv3x3N16.apply(x, w, bias, addend, residual, relu_in, cdt)


class BilinearNHWC(Function):
    @staticmethod
    def forward(ctx, x, ho, wo):
        assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32 and x.shape[3] % 4 == 0
        nb, h, w, c = x.shape
        out = torch.empty((nb, ho, wo, c), dtype=torch.float32, device=x.device)
        rc = lib().ocpg_bilinear_nhwc_fwd(x.data_ptr(), nb, h, w, c, ho, wo, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc:
            check(rc, "ocpg_bilinear_nhwc_fwd")
        ctx.shape = (nb, h, w, c, ho, wo)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        nb, h, w, c, ho, wo = ctx.shape
        if go.dtype != torch.float32 or not go.is_contiguous():
            go = go.float().contiguous()
        gin = torch.empty((nb, h, w, c), dtype=torch.float32, device=go.device)
        rc = lib().ocpg_bilinear_nhwc_bwd(go.data_ptr(), nb, h, w, c, ho, wo, gin.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc:
            check(rc, "ocpg_bilinear_nhwc_bwd")
        return gin, None, None


def bilinear_nhwc(x, size):
    return BilinearNHWC.apply(x, int(size[0]), int(size[1]))
