"""Fused (shifted-)window attention: autograd binding of ocpg_win_attn_{fwd,bwd} (csrc/win_attn.hip).

Replaces the score / bias / mask / softmax / PV chain of the reference's WindowAttention3D.forward
(models/video_swin_transformer.py:138-169); nothing of size N x N is ever written to HBM except the (tiny, per-head)
relative-position bias and its gradient.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib, stream_ptr

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class WindowAttentionFunction(Function):
    @staticmethod
    def forward(ctx, qkv, bias, region, scale, num_windows, bias_t=None):
        """qkv [BW, N, 3, H, 32]; bias [H, N, N] (fp32); region [NW, N] int32 or None -> out [BW, N, H*32]."""
        if not qkv.is_cuda:
            raise RuntimeError("WindowAttentionFunction: qkv must be a GPU tensor: Not implemented on the CPU")
        if qkv.dtype not in _DT:
            raise RuntimeError(f"WindowAttentionFunction: unsupported dtype {qkv.dtype}")
        qkv = qkv.contiguous()
        bw, n, three, h, hd = qkv.shape
        assert three == 3
        bias = bias.float().contiguous()
        if bias_t is None or bias_t.dtype != torch.float32 or not bias_t.is_contiguous() or bias_t.shape != bias.shape:
            bias_t = bias.transpose(1, 2).contiguous()        # (models/video_swin_transformer.py hands the transposed table along: RelPosBias)
        out = torch.empty((bw, n, h * hd), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((bw, h, n), dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            check(lib().ocpg_win_attn_fwd(qkv.data_ptr(), bias_t.data_ptr(), region.data_ptr() if region is not None else None,
                                          float(scale), bw, int(num_windows), n, h, hd, out.data_ptr(), lse.data_ptr(),
                                          _DT[qkv.dtype], stream_ptr()), "ocpg_win_attn_fwd")
        ctx.save_for_backward(qkv, bias, bias_t, out, lse)
        ctx.region, ctx.scale, ctx.num_windows = region, float(scale), int(num_windows)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, bias, bias_t, out, lse = ctx.saved_tensors
        bw, n, _, h, hd = qkv.shape
        dout = dout.to(qkv.dtype).contiguous()
        dqkv = torch.empty_like(qkv)
        dbuf = torch.empty_like(lse)
        region = ctx.region
        if qkv.dtype != torch.float32:
            # matrix-core kernels (csrc/win_attn_mfma.hip): dS leaves as a [BW, H, N, N] tensor in the storage dtype and is summed
            # over the windows here (one streaming reduction instead of BW * H * N^2 float atomics)
            ds = torch.empty((bw, h, n, n), dtype=qkv.dtype, device=qkv.device) if ctx.needs_input_grad[1] else None
            with torch.cuda.device(qkv.device):
                rc = lib().ocpg_win_attn_bwd_mfma(qkv.data_ptr(), bias.data_ptr(), bias_t.data_ptr(),
                                                  region.data_ptr() if region is not None else None, ctx.scale, bw, ctx.num_windows, n, h, hd,
                                                  out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), dbuf.data_ptr(),
                                                  ds.data_ptr() if ds is not None else None, _DT[qkv.dtype], stream_ptr())
            if rc == 0:
                dbias = ds.sum(0, dtype=torch.float32).transpose(1, 2) if ds is not None else None
                return dqkv, dbias, None, None, None, None
            if rc != -2000:
                check(rc, "ocpg_win_attn_bwd_mfma")
        dbias_t = torch.zeros_like(bias_t) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(qkv.device):
            check(lib().ocpg_win_attn_bwd(qkv.data_ptr(), bias.data_ptr(), bias_t.data_ptr(),
                                          region.data_ptr() if region is not None else None, ctx.scale, bw, ctx.num_windows, n, h, hd,
                                          out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), dbuf.data_ptr(),
                                          dbias_t.data_ptr() if dbias_t is not None else None, _DT[qkv.dtype], stream_ptr()),
                  "ocpg_win_attn_bwd")
        dbias = dbias_t.transpose(1, 2) if dbias_t is not None else None
        return dqkv, dbias, None, None, None, None


def window_attention(qkv, bias, region, scale, num_windows, bias_t=None):
    """bias_t (optional, no gradient): bias.transpose(1, 2) already contiguous."""
    return WindowAttentionFunction.apply(qkv, bias, region, scale, num_windows, bias_t)
