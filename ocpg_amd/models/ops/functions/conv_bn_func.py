"""1x1 convolution + frozen-BN affine (+ residual) (+ ReLU) of a channels-last map as ONE autograd node.

Forward: ocpg_gemm (hipBLASLt, cached plan, straight on the NHWC buffers: no view ops) -> ocpg_bn_act_fwd in place.
Backward: ocpg_bn_act_bwd -> input gradient GEMM + weight gradient GEMM (row-split, amp_cache.weight_grad's rule).
Replaces torchvision Bottleneck's conv1+bn1+relu, conv3+bn3(+identity)+relu and downsample conv+bn as the reference
runs them (models/backbone.py:46-56 FrozenBatchNorm2d + nn.Conv2d).  The step is launch-bound on the host: one Python
autograd node and 2 (forward) / 3-4 (backward) C calls replace two nodes and ~10 tensor-view ops per layer.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib

_DT = {torch.float32: 0, torch.bfloat16: 1}
_CL = torch.channels_last
EPILOGUE = os.environ.get("OCPG_GEMM_EPILOGUE", "1") != "0"     # A/B switch: BN affine / skip / ReLU in the GEMM epilogue
# A/B switch: the gradient of a bottleneck's identity skip is ADDED BY THE GEMM that computes conv1's input gradient (C = gz W + 1.0 C on
# the skip gradient's buffer) instead of by autograd's accumulation (one read-read-write pass over the block input per block: 33 per step).
# conv1's forward leaves a token for its input; the block's last convolution finds it when its `skip` IS that input, its backward parks
# the skip gradient in the token and returns None for `skip`; conv1's backward (which runs later) accumulates into the parked buffer.
# A/B switch (default off: measured neutral, 38.8-39.2 vs 39.0 ms): conv3x3_mfma's forward also writes the patch matrix its weight gradient
# contracts with, instead of an im2col launch in the backward (the extra stores cost the forward what the launch cost the backward)
FWD_COLS = os.environ.get("OCPG_CONV3X3_FWD_COLS", "0") != "0"
SKIP_GRAD_IN_GEMM = os.environ.get("OCPG_SKIP_GRAD_IN_GEMM", "1") != "0"
_SKIP_TOKENS = {}           # data_ptr of a conv1 input -> token; cleared at the start of every backbone forward (reset_skip_tokens)
# A/B switch (round 4): a bottleneck's conv1 -> bn1 -> ReLU output has ONE consumer, conv2.  conv2's input-gradient kernel (conv3x3_mfma<DGRAD>)
# then also applies conv1's frozen-BN + ReLU backward in its epilogue (gz1 = gx * scale1 * [y1 > 0]) and conv1's backward skips its
# bn_act_bwd launch (33 launches, ~0.5 ms per step at config #2).  conv1's forward leaves a token under its output's address, conv2's
# forward picks it up when its input IS that tensor.
PREMASK = os.environ.get("OCPG_PREMASK_DGRAD", "1") != "0"
_PREMASK_TOKENS = {}        # data_ptr of a conv1 + bn1 + ReLU output -> token; cleared with the skip tokens


def reset_skip_tokens():
    _SKIP_TOKENS.clear()
    _PREMASK_TOKENS.clear()


def _same_tensor(a, b):
    if a is None or b is None:          # a token whose backward already ran (left behind by a block that was not an identity block)
        return False
    return a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.dtype == b.dtype and a.stride() == b.stride()
                      and a._version == b._version)


def _tensor_key(t):
    return (t.data_ptr(), tuple(t.shape), t.dtype, tuple(t.stride()), t._version)


def _reduce_partials(part, w_is_cast_copy):
    """[S, Co, K] row-split partial products of a weight gradient -> their sum; for the working copy of a parameter (used once per
    forward: every ResNet convolution) the sum is left to the fused gradient cast at the end of the backward (amp_cache.defer_sum)."""
    from ... import amp_cache
    if w_is_cast_copy and amp_cache.DEFER_SUM and amp_cache.MULTI_CAST and part.dtype in (torch.bfloat16, torch.float16):
        return amp_cache.defer_sum(part)
    return part.sum(0)


def eligible(x, conv):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _DT and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.bias is None and x.is_contiguous(memory_format=_CL))


def eligible_s2(x, conv):
    """A 1x1 / stride-2 convolution (the projection shortcut of layer2-4's first block, torchvision resnet.py `downsample`) = the
    stride-1 GEMM on every other pixel of every other row."""
    return (x.is_cuda and x.dim() == 4 and x.dtype in _DT and conv.kernel_size == (1, 1) and conv.stride == (2, 2) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.bias is None and x.is_contiguous(memory_format=_CL))


class Subsample2(Function):
    """x[:, :, ::2, ::2] as a channels-last map; backward: the gradient written into the even pixels of a zeroed map (two launches;
    autograd's own slice backward is two zero-fills + two strided copies)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return x[:, :, ::2, ::2].contiguous(memory_format=_CL)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        gx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device, memory_format=_CL).zero_()
        gx[:, :, ::2, ::2] = g
        return gx


def subsample2(x):
    return Subsample2.apply(x)


class Conv1x1BNAct(Function):
    @staticmethod
    def forward(ctx, x, w, scale, shift, skip, relu, splits):
        n, c, h, wd = x.shape
        co = w.shape[0]
        m = n * h * wd
        dt = _DT[x.dtype]
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        y = torch.empty((n, co, h, wd), dtype=x.dtype, device=x.device, memory_format=_CL)
        if skip is not None and (skip.dtype != x.dtype or not skip.is_contiguous(memory_format=_CL)):
            skip = skip.to(x.dtype).contiguous(memory_format=_CL)
        rc = -1105
        if EPILOGUE:    # y[m, co] = act(scale[co] * x[m, c] w[co, c]^T + shift[co] (+ skip)) inside the GEMM
            rc = L.ocpg_gemm_bn_act(x.data_ptr(), w.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                    None if skip is None else skip.data_ptr(), int(relu), dt, m, co, c, st)
            if rc and rc != -1105:
                check(rc, "ocpg_gemm_bn_act")
        if rc:          # no epilogue kernel for this shape: GEMM, then the frozen-BN kernel in place
            rc = L.ocpg_gemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, dt, dt, 0, 1, m, co, c, c, c, co, 1, 0, 0, 0, 1.0, 0.0, st)
            if rc:
                check(rc, "ocpg_gemm")
            rc = L.ocpg_bn_act_fwd(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None if skip is None else skip.data_ptr(), y.data_ptr(),
                                   m, co, 1, int(relu), dt, st)
            if rc:
                check(rc, "ocpg_bn_act_fwd")
        ctx.save_for_backward(x, w, y, scale)
        from ...amp_cache import is_cast_copy
        ctx.meta = (bool(relu), skip is not None, splits)
        ctx.w_cast = is_cast_copy(w)
        ctx.give = ctx.take = None
        ctx.premask = None
        if PREMASK and relu and skip is None and x.dtype == torch.bfloat16 and ctx.needs_input_grad[1]:
            # (the token identifies y by value, not by reference: ctx -> token -> y -> grad_fn -> ctx would be a cycle that keeps this
            # forward's autograd graph alive until the garbage collector runs -- and a stale graph inside a later stream capture is the
            # hipStreamEndCapture crash of DESIGN.md section 5)
            ctx.premask = {"y": _tensor_key(y), "scale": scale, "gz": None}
            _PREMASK_TOKENS[y.data_ptr()] = ctx.premask
        if SKIP_GRAD_IN_GEMM:
            if skip is not None:
                tok = _SKIP_TOKENS.pop(skip.data_ptr(), None)
                if tok is not None and _same_tensor(tok["x"], skip) and ctx.needs_input_grad[4]:
                    ctx.give = tok                     # this node's skip gradient goes to the token, not to autograd
            elif ctx.needs_input_grad[0]:
                ctx.take = {"x": x, "g": None}
                _SKIP_TOKENS[x.data_ptr()] = ctx.take
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, y, scale = ctx.saved_tensors
        relu, has_skip, splits = ctx.meta
        n, c, h, wd = x.shape
        co = w.shape[0]
        m = n * h * wd
        dt = _DT[x.dtype]
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        if gy.dtype != y.dtype or not gy.is_contiguous(memory_format=_CL):
            gy = gy.to(y.dtype).contiguous(memory_format=_CL)
        need_x, need_w, need_skip = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_skip and ctx.needs_input_grad[4]
        tok = ctx.premask
        if tok is not None and tok["gz"] is not None and tok["gz"].data_ptr() == gy.data_ptr() and tok["gz"].shape == gy.shape:
            gz, gskip = gy, None            # the consumer's input-gradient kernel already applied this layer's frozen-BN + ReLU backward
            tok["gz"] = None
        else:
            gz = torch.empty_like(y)
            gskip = torch.empty_like(y) if need_skip else None
            rc = L.ocpg_bn_act_bwd(gy.data_ptr(), y.data_ptr(), scale.data_ptr(), gz.data_ptr(), None if gskip is None else gskip.data_ptr(),
                                   m, co, 1, int(relu), dt, st)
            if rc:
                check(rc, "ocpg_bn_act_bwd")
        gx = gw = None
        if ctx.give is not None and need_skip:
            ctx.give["g"] = gskip                      # parked for conv1's input-gradient GEMM of this block (runs later in this backward)
            gskip = None
        parked = None
        if ctx.take is not None:
            parked, ctx.take["g"], ctx.take["x"] = ctx.take["g"], None, None
        if need_x:      # gx[m, c] = gz[m, co] w[co, c] (+ the block's skip gradient, accumulated in place)
            acc = parked is not None and parked.dtype == x.dtype and parked.shape == x.shape and parked.is_contiguous(memory_format=_CL)
            gx = parked if acc else torch.empty((n, c, h, wd), dtype=x.dtype, device=x.device, memory_format=_CL)
            rc = L.ocpg_gemm(gz.data_ptr(), w.data_ptr(), gx.data_ptr(), None, dt, dt, 0, 0, m, c, co, co, c, c, 1, 0, 0, 0, 1.0,
                             1.0 if acc else 0.0, st)
            if rc:
                check(rc, "ocpg_gemm")
            if parked is not None and not acc:
                gx = gx + parked
        elif parked is not None:
            gx = parked
        if need_w:      # gw[co, c] = gz[m, co]^T x[m, c], rows split into `splits` chunks (one strided-batched GEMM + a sum)
            from ...amp_cache import side_wgrad
            with side_wgrad(ctx.w_cast, gz, x) as sw:          # off the critical path: the weight-gradient stream (amp_cache.side_wgrad)
                st = torch.cuda.current_stream().cuda_stream
                if splits > 1 and m % splits == 0:
                    r = m // splits
                    part = torch.empty((splits, co, c), dtype=x.dtype, device=x.device)
                    rc = L.ocpg_gemm(gz.data_ptr(), x.data_ptr(), part.data_ptr(), None, dt, dt, 1, 0, co, c, r, co, c, c, splits, r * co, r * c,
                                     co * c, 1.0, 0.0, st)
                    gw = _reduce_partials(part, ctx.w_cast)
                else:
                    gw = torch.empty((co, c), dtype=x.dtype, device=x.device)
                    rc = L.ocpg_gemm(gz.data_ptr(), x.data_ptr(), gw.data_ptr(), None, dt, dt, 1, 0, co, c, m, co, c, c, 1, 0, 0, 0, 1.0, 0.0, st)
                if rc:
                    check(rc, "ocpg_gemm")
                gw = sw.publish(gw).view(w.shape)
        return gx, gw, None, None, gskip, None, None


def conv1x1_bn_act(x, w, scale, shift, skip, relu, splits):
    return Conv1x1BNAct.apply(x, w, scale, shift, skip, relu, splits)


# ---- 3x3 conv (padding == dilation, stride 1|2) + frozen BN + ReLU: im2col (HIP) -> GEMM with the BN/ReLU epilogue ---------
_DT3 = {torch.float32: 0, torch.bfloat16: 1}


def eligible3x3(x, conv):
    """Where the patch-matrix GEMM beats MIOpen (tools/bench_conv3x3.py): >= 256 channels on maps of <= 12288 output pixels."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in _DT3 and conv.kernel_size == (3, 3) and conv.groups == 1 and conv.bias is None
            and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2) and conv.dilation[0] == conv.dilation[1]
            and conv.padding == conv.dilation and conv.padding_mode == "zeros" and x.is_contiguous(memory_format=_CL)):
        return False
    s = conv.stride[0]
    rows = x.shape[0] * ((x.shape[2] - 1) // s + 1) * ((x.shape[3] - 1) // s + 1)
    return x.shape[1] >= 256 and (x.shape[1] * x.element_size()) % 16 == 0 and rows <= 12288


class Conv3x3BNAct(Function):
    @staticmethod
    def forward(ctx, x, w, scale, shift, relu, stride, dil, splits):
        n, c, h, wd = x.shape
        co = w.shape[0]
        ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
        m = n * ho * wo
        dt = _DT3[x.dtype]
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        cols = torch.empty((m, 9 * c), dtype=x.dtype, device=x.device)
        check(L.ocpg_im2col3x3_nhwc(x.data_ptr(), n, h, wd, c, stride, dil, cols.data_ptr(), dt, st), "ocpg_im2col3x3_nhwc")
        w2 = w.permute(0, 2, 3, 1).reshape(co, 9 * c)             # a view when the weight is channels-last
        if not w2.is_contiguous():
            w2 = w2.contiguous()
        y = torch.empty((n, co, ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)
        rc = L.ocpg_gemm_bn_act(cols.data_ptr(), w2.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, int(relu), dt, m, co,
                                9 * c, st) if EPILOGUE else -1105
        if rc and rc != -1105:
            check(rc, "ocpg_gemm_bn_act")
        if rc:
            check(L.ocpg_gemm(cols.data_ptr(), w2.data_ptr(), y.data_ptr(), None, dt, dt, 0, 1, m, co, 9 * c, 9 * c, 9 * c, co, 1, 0, 0, 0, 1.0,
                              0.0, st), "ocpg_gemm")
            check(L.ocpg_bn_act_fwd(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, y.data_ptr(), m, co, 1, int(relu), dt, st),
                  "ocpg_bn_act_fwd")
        ctx.save_for_backward(cols, w2, y, scale)
        ctx.meta = (bool(relu), splits, (n, c, h, wd, ho, wo, stride, dil), w.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        cols, w2, y, scale = ctx.saved_tensors
        relu, splits, (n, c, h, wd, ho, wo, stride, dil), wshape = ctx.meta
        co = w2.shape[0]
        m, k = cols.shape
        dt = _DT3[y.dtype]
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        if gy.dtype != y.dtype or not gy.is_contiguous(memory_format=_CL):
            gy = gy.to(y.dtype).contiguous(memory_format=_CL)
        gz = torch.empty_like(y)
        check(L.ocpg_bn_act_bwd(gy.data_ptr(), y.data_ptr(), scale.data_ptr(), gz.data_ptr(), None, m, co, 1, int(relu), dt, st), "ocpg_bn_act_bwd")
        gx = gw = None
        if ctx.needs_input_grad[0]:     # dcols[m, 9c] = gz[m, co] w2[co, 9c] ; gx = col2im(dcols)
            dcols = torch.empty((m, k), dtype=y.dtype, device=y.device)
            check(L.ocpg_gemm(gz.data_ptr(), w2.data_ptr(), dcols.data_ptr(), None, dt, dt, 0, 0, m, k, co, co, k, k, 1, 0, 0, 0, 1.0, 0.0, st),
                  "ocpg_gemm")
            gx = torch.empty((n, c, h, wd), dtype=y.dtype, device=y.device, memory_format=_CL)
            check(L.ocpg_col2im3x3_nhwc(dcols.data_ptr(), n, h, wd, c, stride, dil, gx.data_ptr(), dt, st), "ocpg_col2im3x3_nhwc")
        if ctx.needs_input_grad[1]:     # gw[co, 9c] = gz^T cols, rows split
            if splits > 1 and m % splits == 0:
                r = m // splits
                part = torch.empty((splits, co, k), dtype=y.dtype, device=y.device)
                check(L.ocpg_gemm(gz.data_ptr(), cols.data_ptr(), part.data_ptr(), None, dt, dt, 1, 0, co, k, r, co, k, k, splits, r * co, r * k,
                                  co * k, 1.0, 0.0, st), "ocpg_gemm")
                g2 = part.sum(0)
            else:
                g2 = torch.empty((co, k), dtype=y.dtype, device=y.device)
                check(L.ocpg_gemm(gz.data_ptr(), cols.data_ptr(), g2.data_ptr(), None, dt, dt, 1, 0, co, k, m, co, k, k, 1, 0, 0, 0, 1.0, 0.0, st),
                      "ocpg_gemm")
            gw = g2.view(co, 3, 3, c).permute(0, 3, 1, 2)          # channels-last strides of [co, c, 3, 3]
        return gx, gw, None, None, None, None, None, None


def conv3x3_bn_act(x, w, scale, shift, relu, stride, dil, splits):
    return Conv3x3BNAct.apply(x, w, scale, shift, relu, int(stride), int(dil), splits)


# ---- 3x3 conv (padding 1, stride 1|2) + frozen BN + ReLU on the matrix cores: csrc/conv3x3_mfma.hip ------------------------
def eligible3x3_mfma(x, conv):
    """bf16 channels-last map, 3x3 / padding 1 / dilation 1 / stride 1|2 / no bias, channel counts the kernel's K step and its
    output tile serve (multiples of 64, >= _MFMA_MIN_C: ResNet layers 2-4; the 64-channel layer1 keeps MIOpen, which is faster there)."""
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and conv.kernel_size == (3, 3) and conv.groups == 1
            and conv.bias is None and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2) and conv.dilation == (1, 1)
            and conv.padding == (1, 1) and conv.padding_mode == "zeros" and x.is_contiguous(memory_format=_CL)
            and x.shape[1] % 64 == 0 and conv.out_channels % 64 == 0 and x.shape[1] >= _MFMA_MIN_C and conv.out_channels >= _MFMA_MIN_C)


BODY_SPLITK = os.environ.get("OCPG_CONV3X3_SPLITK", "0") != "0"     # opt-in (-1 = where the tile grid is small, n = force n): split-K forward / input gradient of the body's 3x3 convs; measured +0.45 ms per step at 2 clips, neutral at 1 (r4)
WGRAD_OWN = os.environ.get("OCPG_WGRAD_OWN", "1") != "0"     # A/B switch: conv3x3_mfma's weight gradient by csrc/conv3x3_wgrad.hip (0 = im2col + row-split GEMM)
DGRAD_OWN_WEIGHT = os.environ.get("OCPG_DGRAD_OWN_WEIGHT", "1") != "0"     # A/B switch: conv3x3_mfma's input gradient reads the weight untransposed
_MFMA_MIN_C = int(os.environ.get("OCPG_MFMA_CONV3X3_MIN_C", "128"))     # 64 also serves layer1 (frozen: forward only), measured 0.08 ms/step SLOWER than MIOpen there (r4)


class Conv3x3MfmaBNAct(Function):
    """y = act(bn(conv3x3(x))): ONE launch forward (implicit-GEMM MFMA kernel with the frozen-BN affine + ReLU in its epilogue).
    Backward: frozen-BN/ReLU backward (HIP) -> input gradient by the same MFMA kernel on the channel-swapped weight ->
    weight gradient as im2col (HIP) + hipBLASLt GEMM over the rows (split for occupancy)."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, relu, stride, splits):
        n, c, h, wd = x.shape
        co = w.shape[0]
        ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
        w2 = w.permute(0, 2, 3, 1)                                   # [co,3,3,c]: a view when the weight is channels-last
        if not w2.is_contiguous():
            w2 = w2.contiguous()
        y = torch.empty((n, co, ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)
        st = torch.cuda.current_stream().cuda_stream
        # the forward kernel gathers exactly the rows of the patch matrix the weight gradient needs: kept (44 MB per layer3 conv, 1.4 GB over
        # the ResNet-101 body at 10 frames) instead of re-gathered by an im2col launch in the backward
        cols = torch.empty((n * ho * wo, 9 * c), dtype=x.dtype, device=x.device) if (FWD_COLS and ctx.needs_input_grad[1]) else None
        sp = int(lib().ocpg_conv3x3_mfma_body_splits(n * ho * wo, co, c)) if (BODY_SPLITK and cols is None) else 1
        if sp > 1:      # few tiles, long K: the K chain split over the grid, the BN + ReLU epilogue in the summing pass (csrc/conv3x3_mfma.hip)
            part = torch.empty((sp, n * ho * wo, co), dtype=torch.float32, device=x.device)
            check(lib().ocpg_conv3x3_mfma_fwd_bn_splitk(x.data_ptr(), w2.data_ptr(), scale.data_ptr(), shift.data_ptr(), int(relu), n, h, wd, c, co, stride,
                                                        sp, part.data_ptr(), y.data_ptr(), st), "ocpg_conv3x3_mfma_fwd_bn_splitk")
        else:
            check(lib().ocpg_conv3x3_mfma_fwd_cols(x.data_ptr(), w2.data_ptr(), scale.data_ptr(), shift.data_ptr(), int(relu), n, h, wd, c, co, stride,
                                                   y.data_ptr(), None if cols is None else cols.data_ptr(), st), "ocpg_conv3x3_mfma_fwd_cols")
        ctx.has_cols = cols is not None
        if cols is not None:
            ctx.save_for_backward(x, w2, y, scale, cols)
        else:
            ctx.save_for_backward(x, w2, y, scale)
        from ...amp_cache import is_cast_copy
        ctx.meta = (bool(relu), splits, stride)
        ctx.w_cast = is_cast_copy(w)
        tok = _PREMASK_TOKENS.pop(x.data_ptr(), None) if PREMASK else None
        ctx.premask = tok if (tok is not None and tok["y"] == _tensor_key(x) and ctx.needs_input_grad[0]) else None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w2, y, scale = ctx.saved_tensors[:4]
        relu, splits, stride = ctx.meta
        n, c, h, wd = x.shape
        co, ho, wo = y.shape[1], y.shape[2], y.shape[3]
        m, k = n * ho * wo, 9 * c
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        if gy.dtype != y.dtype or not gy.is_contiguous(memory_format=_CL):
            gy = gy.to(y.dtype).contiguous(memory_format=_CL)
        gz = torch.empty_like(y)
        check(L.ocpg_bn_act_bwd(gy.data_ptr(), y.data_ptr(), scale.data_ptr(), gz.data_ptr(), None, m, co, 1, int(relu), 1, st), "ocpg_bn_act_bwd")
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((n, c, h, wd), dtype=y.dtype, device=y.device, memory_format=_CL)
            tok = ctx.premask       # x IS the layer in front's bn + ReLU output: its backward rides in this kernel's epilogue
            mask_ptr, scale_ptr = (x.data_ptr(), tok["scale"].data_ptr()) if tok is not None else (None, None)
            sp = int(L.ocpg_conv3x3_mfma_body_splits(n * h * wd, c, co)) if (BODY_SPLITK and DGRAD_OWN_WEIGHT and c % 8 == 0) else 1
            if sp > 1:
                part = torch.empty((sp, n * h * wd, c), dtype=torch.float32, device=y.device)
                check(L.ocpg_conv3x3_mfma_dgrad_w_splitk(gz.data_ptr(), w2.data_ptr(), mask_ptr, scale_ptr, n, h, wd, c, co, stride, sp, part.data_ptr(),
                                                         gx.data_ptr(), st), "ocpg_conv3x3_mfma_dgrad_w_splitk")
            elif DGRAD_OWN_WEIGHT and c % 8 == 0:      # the weight as it lies ([co,3,3,c]): transposing LDS reads, no per-step transposed copy
                check(L.ocpg_conv3x3_mfma_dgrad_w(gz.data_ptr(), w2.data_ptr(), mask_ptr, scale_ptr, n, h, wd, c, co, stride, gx.data_ptr(), st),
                      "ocpg_conv3x3_mfma_dgrad_w")
            else:
                wt = w2.permute(3, 1, 2, 0).contiguous()               # [c,3,3,co]
                check(L.ocpg_conv3x3_mfma_dgrad_masked(gz.data_ptr(), wt.data_ptr(), mask_ptr, scale_ptr, n, h, wd, c, co, stride, gx.data_ptr(), st),
                      "ocpg_conv3x3_mfma_dgrad_masked")
            if tok is not None:
                tok["gz"] = gx
        if ctx.needs_input_grad[1]:
            from ...amp_cache import side_wgrad
            with side_wgrad(ctx.w_cast, gz, x) as sw:          # off the critical path: the weight-gradient stream (amp_cache.side_wgrad)
                st = torch.cuda.current_stream().cuda_stream
                if WGRAD_OWN and not ctx.has_cols and c % 8 == 0 and co % 8 == 0:
                    # straight from the two maps (csrc/conv3x3_wgrad.hip): no patch matrix, no library GEMM
                    sp = int(L.ocpg_conv3x3_mfma_wgrad_splits(n, h, wd, c, co, stride))
                    part = torch.empty((sp, co, k), dtype=y.dtype, device=y.device)
                    check(L.ocpg_conv3x3_mfma_wgrad(gz.data_ptr(), x.data_ptr(), n, h, wd, c, co, stride, part.data_ptr(), st), "ocpg_conv3x3_mfma_wgrad")
                    g2 = _reduce_partials(part, ctx.w_cast) if sp > 1 else part[0]
                else:
                    if ctx.has_cols:
                        cols = ctx.saved_tensors[4]
                    else:
                        cols = torch.empty((m, k), dtype=y.dtype, device=y.device)
                        check(L.ocpg_im2col3x3_nhwc(x.data_ptr(), n, h, wd, c, stride, 1, cols.data_ptr(), 1, st), "ocpg_im2col3x3_nhwc")
                    if splits > 1 and m % splits == 0:
                        r = m // splits
                        part = torch.empty((splits, co, k), dtype=y.dtype, device=y.device)
                        check(L.ocpg_gemm(gz.data_ptr(), cols.data_ptr(), part.data_ptr(), None, 1, 1, 1, 0, co, k, r, co, k, k, splits, r * co, r * k,
                                          co * k, 1.0, 0.0, st), "ocpg_gemm")
                        g2 = _reduce_partials(part, ctx.w_cast)
                    else:
                        g2 = torch.empty((co, k), dtype=y.dtype, device=y.device)
                        check(L.ocpg_gemm(gz.data_ptr(), cols.data_ptr(), g2.data_ptr(), None, 1, 1, 1, 0, co, k, m, co, k, k, 1, 0, 0, 0, 1.0, 0.0, st),
                              "ocpg_gemm")
                gw = sw.publish(g2).view(co, 3, 3, c).permute(0, 3, 1, 2)          # channels-last strides of [co, c, 3, 3]
        return gx, gw, None, None, None, None, None


def conv3x3_mfma_bn_act(x, w, scale, shift, relu, stride, splits):
    return Conv3x3MfmaBNAct.apply(x, w, scale, shift, relu, int(stride), splits)


# ---- 3x3 conv (padding 1, stride 1|2) with bias, few output pixels and a long reduction: split-K on the matrix cores -------------------------
def eligible3x3_splitk(x, conv):
    """The neck's extra level (models/ocpg.py:119-123): bf16 channels-last map, 3x3 / padding 1 / stride 1|2, channel counts the kernel's
    64-wide K step and output tile serve, and a GEMM so short in rows that the plain kernel would leave the chip empty
    (ocpg_conv3x3_mfma_splits > 1)."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and conv.kernel_size == (3, 3) and conv.groups == 1
            and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2) and conv.dilation == (1, 1) and conv.padding == (1, 1)
            and conv.padding_mode == "zeros" and x.is_contiguous(memory_format=_CL) and x.shape[1] % 64 == 0 and conv.out_channels % 64 == 0):
        return False
    n, c, h, w = x.shape
    return int(lib().ocpg_conv3x3_mfma_splits(n, h, w, c, conv.out_channels, conv.stride[0])) > 1


class Conv3x3SplitK(Function):
    """y = conv3x3(x, w) + b with K split over the grid (csrc/conv3x3_mfma.hip, SPLITK) + a summing pass.  The forward keeps the patch
    matrix it gathered (the weight gradient's operand: gy^T cols, one hipBLASLt GEMM over the few rows); the input gradient is the MFMA
    kernel on the channel-swapped weight (its GEMM has N*H*W rows: no split)."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        n, c, h, wd = x.shape
        co = w.shape[0]
        ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
        m = n * ho * wo
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        w2 = w.permute(0, 2, 3, 1)                                   # [co,3,3,c]: a view when the weight is channels-last
        if not w2.is_contiguous():
            w2 = w2.contiguous()
        splits = int(L.ocpg_conv3x3_mfma_splits(n, h, wd, c, co, stride))
        part = torch.empty((splits, m, co), dtype=torch.float32, device=x.device)
        y = torch.empty((n, co, ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)
        need_w = ctx.needs_input_grad[1]
        cols = torch.empty((m, 9 * c), dtype=x.dtype, device=x.device) if need_w else None
        bf = None if b is None else b.float()
        check(L.ocpg_conv3x3_mfma_fwd_splitk(x.data_ptr(), w2.data_ptr(), None if bf is None else bf.data_ptr(), n, h, wd, c, co, stride, splits,
                                             part.data_ptr(), y.data_ptr(), 1, None if cols is None else cols.data_ptr(), st),
              "ocpg_conv3x3_mfma_fwd_splitk")
        ctx.save_for_backward(w2, cols if need_w else w2)
        ctx.meta = (n, c, h, wd, ho, wo, stride, b is not None, need_w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        w2, cols = ctx.saved_tensors
        n, c, h, wd, ho, wo, stride, has_b, need_w = ctx.meta
        co = w2.shape[0]
        m, k = n * ho * wo, 9 * c
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        if gy.dtype != torch.bfloat16 or not gy.is_contiguous(memory_format=_CL):
            gy = gy.to(torch.bfloat16).contiguous(memory_format=_CL)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = w2.permute(3, 1, 2, 0).contiguous()               # [c,3,3,co]
            gx = torch.empty((n, c, h, wd), dtype=gy.dtype, device=gy.device, memory_format=_CL)
            check(L.ocpg_conv3x3_mfma_dgrad(gy.data_ptr(), wt.data_ptr(), n, h, wd, c, co, stride, gx.data_ptr(), st), "ocpg_conv3x3_mfma_dgrad")
        if need_w and ctx.needs_input_grad[1]:
            g2 = torch.empty((co, k), dtype=gy.dtype, device=gy.device)
            check(L.ocpg_gemm(gy.data_ptr(), cols.data_ptr(), g2.data_ptr(), None, 1, 1, 1, 0, co, k, m, co, k, k, 1, 0, 0, 0, 1.0, 0.0, st), "ocpg_gemm")
            gw = g2.view(co, 3, 3, c).permute(0, 3, 1, 2)          # channels-last strides of [co, c, 3, 3]
        if has_b and ctx.needs_input_grad[2]:
            gb = gy.permute(0, 2, 3, 1).reshape(m, co).float().sum(0).to(gy.dtype)
        return gx, gw, gb, None


def conv3x3_splitk(x, w, b, stride):
    return Conv3x3SplitK.apply(x, w, b, int(stride))
