"""Autograd binding of csrc/attn_smallk.hip: multi-head attention against a short key sequence (the vision-language fusion
gate, models/segmentation.py:95-113, and the decoder's 5-query self-attention, models/deformable_transformer.py:323-326).

`attention(q, k, v, key_padding_mask, scale, H, pdrop)` takes the projections' outputs as they are ([L, B, C] rows, any row
stride) and returns [Lq, B, C]; it returns None when the HIP kernel does not serve the shape (head_dim != 32, more than 32
keys, ...) so that the caller keeps its generic path -- there is no silent fallback INSIDE this module.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib, stream_ptr
from .fused_ln_func import _unpack

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _rows(t):
    """[L, B, C] tensor whose rows (l, b) sit at (l * B + b) * ld: returns (tensor, ld), copying only if the view is not of that form."""
    if t.stride(-1) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)) or t.stride(1) < t.shape[2]:
        t = t.contiguous()
    return t, t.stride(1)


def supported(q, k, H, pdrop):
    C = q.shape[-1]
    return (q.is_cuda and q.dtype in _DT and C % H == 0 and C // H == 32 and H <= 8 and 256 % H == 0 and k.shape[0] <= 32
            and q.shape[1] <= 65535 and 0.0 <= pdrop < 1.0)


class SmallKeyAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v, key_pad, scale, H, pdrop, rng):
        q, ldq = _rows(q)
        k, ldk = _rows(k)
        v, ldv = _rows(v)
        Lq, B, C = q.shape
        Lk = k.shape[0]
        pad = None if key_pad is None else key_pad.to(torch.uint8).contiguous()
        out = torch.empty((Lq, B, C), dtype=q.dtype, device=q.device)
        lse = torch.empty((Lq, B, H), dtype=torch.float32, device=q.device)
        seed, offset, base = (0, 0, None) if pdrop <= 0 else _unpack(rng)
        with torch.cuda.device(q.device):
            rc = lib().ocpg_attn_smallk_fwd(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, None if pad is None else pad.data_ptr(),
                                            float(scale), Lq, B, H, C // H, Lk, float(pdrop), seed, offset, base, out.data_ptr(), C,
                                            lse.data_ptr(), _DT[q.dtype], stream_ptr())
        check(rc, "ocpg_attn_smallk_fwd")
        ctx.save_for_backward(q, k, v, pad, lse)
        ctx.meta = (float(scale), H, float(pdrop), seed, offset, base, ldq, ldk, ldv)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v, pad, lse = ctx.saved_tensors
        scale, H, pdrop, seed, offset, base, ldq, ldk, ldv = ctx.meta
        Lq, B, C = q.shape
        Lk = k.shape[0]
        dout = dout.contiguous()
        dq = torch.empty((Lq, B, C), dtype=q.dtype, device=q.device)
        dkv = torch.zeros((2, Lk, B, C), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            rc = lib().ocpg_attn_smallk_bwd(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, None if pad is None else pad.data_ptr(),
                                            dout.data_ptr(), C, lse.data_ptr(), scale, Lq, B, H, C // H, Lk, pdrop, seed, offset, base,
                                            dq.data_ptr(), C, dkv[0].data_ptr(), dkv[1].data_ptr(), _DT[q.dtype], stream_ptr())
        check(rc, "ocpg_attn_smallk_bwd")
        dkv = dkv.to(k.dtype)
        return dq, dkv[0], dkv[1], None, None, None, None, None


def attention(q, k, v, key_padding_mask, scale, H, pdrop=0.0, rng=None):
    if not supported(q, k, H, pdrop):
        return None
    return SmallKeyAttention.apply(q, k, v, key_padding_mask, scale, H, pdrop, rng)


class SmallKeyAttentionBF(Function):
    """The same attention for queries stored BATCH-FIRST, q [B, Lq, C] (round 4: the visual tokens of the text gate are the channels-last
    feature map itself, [b, (t h w), c]; putting them token-major and back cost two transposing copies of the map each way).  One launch
    per batch entry on the [L, 1, C] view of its rows; k / v stay [Lk, B, C] (row stride B * C per entry, no copy).  No dropout."""

    @staticmethod
    def forward(ctx, q, k, v, key_pad, scale, H):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, Lq, C = q.shape
        Lk = k.shape[0]
        pad = None if key_pad is None else key_pad.to(torch.uint8).contiguous()
        out = torch.empty((B, Lq, C), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, Lq, H), dtype=torch.float32, device=q.device)
        es = q.element_size()
        with torch.cuda.device(q.device):
            for b in range(B):
                rc = lib().ocpg_attn_smallk_fwd(q.data_ptr() + b * Lq * C * es, C, k.data_ptr() + b * C * es, B * C, v.data_ptr() + b * C * es, B * C,
                                                None if pad is None else pad.data_ptr() + b * Lk, float(scale), Lq, 1, H, C // H, Lk, 0.0, 0, 0, None,
                                                out.data_ptr() + b * Lq * C * es, C, lse.data_ptr() + b * Lq * H * 4, _DT[q.dtype], stream_ptr())
                check(rc, "ocpg_attn_smallk_fwd")
        ctx.save_for_backward(q, k, v, pad, lse)
        ctx.meta = (float(scale), H)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v, pad, lse = ctx.saved_tensors
        scale, H = ctx.meta
        B, Lq, C = q.shape
        Lk = k.shape[0]
        dout = dout.contiguous()
        dq = torch.empty((B, Lq, C), dtype=q.dtype, device=q.device)
        dkv = torch.zeros((2, B, Lk, C), dtype=torch.float32, device=q.device)
        es = q.element_size()
        with torch.cuda.device(q.device):
            for b in range(B):
                rc = lib().ocpg_attn_smallk_bwd(q.data_ptr() + b * Lq * C * es, C, k.data_ptr() + b * C * es, B * C, v.data_ptr() + b * C * es, B * C,
                                                None if pad is None else pad.data_ptr() + b * Lk, dout.data_ptr() + b * Lq * C * es, C,
                                                lse.data_ptr() + b * Lq * H * 4, scale, Lq, 1, H, C // H, Lk, 0.0, 0, 0, None,
                                                dq.data_ptr() + b * Lq * C * es, C, dkv[0, b].data_ptr(), dkv[1, b].data_ptr(), _DT[q.dtype],
                                                stream_ptr())
                check(rc, "ocpg_attn_smallk_bwd")
        dkv = dkv.transpose(1, 2).to(k.dtype)               # [2, Lk, B, C]
        return dq, dkv[0], dkv[1], None, None, None


def attention_batch_first(q, k, v, key_padding_mask, scale, H):
    """q [B, Lq, C], k / v [Lk, B, C] -> [B, Lq, C]; None when csrc/attn_smallk.hip does not serve the shape."""
    if not (q.is_cuda and q.dtype in _DT and q.shape[-1] % H == 0 and q.shape[-1] // H == 32 and H <= 8 and 256 % H == 0 and k.shape[0] <= 32):
        return None
    return SmallKeyAttentionBF.apply(q, k, v, key_padding_mask, scale, H)
