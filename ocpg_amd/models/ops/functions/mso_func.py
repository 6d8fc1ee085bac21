"""MSO's convolutions and its x2 resize on channels-last maps (csrc/mso.hip) as autograd nodes.

conv3x3_n16(x, w, ...) = conv2d(relu?(x), w, padding=1) (+ bias) (+ addend broadcast over groups of images) (+ residual) with
x [NB, H, W, C], w [co <= 16, 9, C] (tap-major: the reference's [co, C, 3, 3] weight permuted to (0, 2, 3, 1)), fp32 output
[NB, H, W, co].  Reference: models/decoder.py:22-46 (`conv1_1div8/4`, `conv2_1div8/4`, `out_conv`, F.interpolate).
"""
import os

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


WGRAD_ATOMIC = os.environ.get("OCPG_MSO_WGRAD_ATOMIC", "1") != "0"    # A/B switch: bands add into the weight gradient (no reduction pass)


def _dgrad(go, w, mask, residual, out_dtype, cdt):
    """Input gradient of conv(act(x), w): the same convolution with flipped taps and the channel axes swapped (+ ReLU'(x) as the
    output mask, + a gradient arriving over a skip connection)."""
    co, _, c = w.shape
    wt = w.view(co, 3, 3, c).flip(1, 2).permute(3, 1, 2, 0).reshape(c, 9, co).to(_TORCH[cdt])      # .to / reshape: one contiguous copy
    if not wt.is_contiguous():
        wt = wt.contiguous()
    return _conv(go, False, wt, None, None, mask, residual, out_dtype, cdt)


def _wgrad(x, relu_in, go, w, want_bias, cdt):
    """(weight gradient in w's dtype and layout, bias gradient | None) of conv(act(x), w) + bias."""
    nb, h, wd, c = x.shape
    co = w.shape[0]
    L = lib()
    rows = int(L.ocpg_mso_wgrad_rows(nb, h, c, cdt))
    n = co * 9 * c
    if WGRAD_ATOMIC:
        buf = torch.zeros(n + 16, dtype=torch.float32, device=x.device)
        part, part_b = buf[:n], buf[n:]
    else:
        bands = nb * ((h + rows - 1) // rows)
        part = torch.empty((bands, n), dtype=torch.float32, device=x.device)
        part_b = torch.empty((bands, 16), dtype=torch.float32, device=x.device) if want_bias else None
    rc = L.ocpg_mso_wgrad(x.data_ptr(), _DT[x.dtype], int(relu_in), go.data_ptr(), part.data_ptr(), _ptr(part_b) if want_bias else None,
                          int(WGRAD_ATOMIC), nb, h, wd, c, co, rows, cdt, torch.cuda.current_stream().cuda_stream)
    if rc:
        check(rc, "ocpg_mso_wgrad")
    if not WGRAD_ATOMIC:
        part = part.sum(0)
        part_b = part_b.sum(0) if want_bias else None
    return part.view(co, 9, c).to(w.dtype), (part_b[:co] if want_bias else None)


def _f32c(go):
    return go if go.dtype == torch.float32 and go.is_contiguous() else go.float().contiguous()


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
        go = _f32c(go)
        gx = gw = gb = ga = None
        if ctx.needs_input_grad[0]:
            gx = _dgrad(go, w, x if relu_in else None, None, x.dtype, cdt)
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw, gb = _wgrad(x, relu_in, go, w, want_b, cdt)        # the band sums of g ride along in the weight-gradient kernel
        elif want_b:
            gb = go.view(-1, co).sum(0)
        if na is not None and ctx.needs_input_grad[3]:
            ga = go if na == nb else go.view(nb // na, na, h, wd, co).sum(0)
        return gx, gw, gb, ga, (go if has_res and ctx.needs_input_grad[4] else None), None, None


def conv3x3_n16(x, w, bias=None, addend=None, residual=None, relu_in=False, cdt=0):
    return Conv3x3N16.apply(x, w, bias, addend, residual, relu_in, cdt)


class ResBlockN16(Function):
    """out = p + conv(relu(conv(relu(p), wm) + shared[n % NA]), w2) + b2: one scale of MSO's mask path (decoder.py:34-35 / 41-42 with
    the feature half of the first convolution precomputed as `shared`).  As ONE node the skip connection's gradient is added in the
    epilogue of the first convolution's input-gradient kernel instead of by autograd's accumulation (two full-map adds per scale)."""

    @staticmethod
    def forward(ctx, p, shared, wm, w2, b2, cdt):
        assert p.dim() == 4 and p.is_contiguous() and p.dtype == torch.float32 and shared.dtype == torch.float32 and shared.is_contiguous()
        assert p.shape[0] % shared.shape[0] == 0 and tuple(shared.shape[1:]) == tuple(p.shape[1:]) and b2.dtype == torch.float32
        for w in (wm, w2):
            assert w.is_contiguous() and tuple(w.shape) == (p.shape[3], 9, p.shape[3]) and w.dtype in _DT
        y = _conv(p, True, wm, None, shared, None, None, torch.float32, cdt)
        out = _conv(y, True, w2, b2, None, None, p, torch.float32, cdt)
        ctx.save_for_backward(p, y, wm, w2)
        ctx.meta = (cdt, shared.shape[0])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        p, y, wm, w2 = ctx.saved_tensors
        cdt, na = ctx.meta
        nb, h, wd, c = p.shape
        go = _f32c(go)
        need = ctx.needs_input_grad
        gy = _dgrad(go, w2, y, None, torch.float32, cdt)
        gw2, gb2 = _wgrad(y, True, go, w2, need[4], cdt) if need[3] else (None, go.view(-1, c).sum(0) if need[4] else None)
        gp = _dgrad(gy, wm, p, go, torch.float32, cdt) if need[0] else None
        gwm = _wgrad(p, True, gy, wm, False, cdt)[0] if need[2] else None
        gs = None
        if need[1]:
            gs = gy if na == nb else gy.view(nb // na, na, h, wd, c).sum(0)
        return gp, gs, gwm, gw2, gb2, None


def res_block_n16(p, shared, wm, w2, b2, cdt=0):
    return ResBlockN16.apply(p, shared, wm, w2, b2, cdt)


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
