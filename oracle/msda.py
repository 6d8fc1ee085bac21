"""Oracle for the MSDeformAttn op (TEST INFRASTRUCTURE) -- two independent CPU restatements.

* ``msda_torch``: stock-torch restatement of the reference's pure-PyTorch ground truth
  ``ms_deform_attn_core_pytorch`` (models/ops/functions/ms_deform_attn_func.py:41-61): per level,
  ``F.grid_sample(bilinear, zeros, align_corners=False)`` of the head-major value map at
  ``2*loc-1``, weighted by the attention weights and summed over (level, point).  Differentiable
  through autograd (this is how the reference's test.py gets its gradient ground truth).
* ``msda_c_forward / msda_c_backward``: the plain-C restatement of the CUDA kernels' arithmetic
  (oracle/msda_oracle.c), fp32 and fp64.
Both are pinned against tests/golden/msda_testpy.npz / msda_cases.npz (tests/test_oracle_msda.py).
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from . import build as _build


def msda_torch(value, spatial_shapes, sampling_locations, attention_weights):
    n, _, m, d = value.shape
    _, lq, _, nl, p, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    per_level = value.split(sizes, dim=1)
    grids = 2.0 * sampling_locations - 1.0
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        h, w = int(h), int(w)
        vmap = per_level[lvl].permute(0, 2, 3, 1).reshape(n * m, d, h, w)          # [N*M, D, H, W]
        g = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(n * m, lq, p, 2)     # [N*M, Lq, P, 2]
        sampled.append(F.grid_sample(vmap, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    sampled = torch.stack(sampled, dim=-2).reshape(n * m, d, lq, nl * p)            # [N*M, D, Lq, L*P]
    w_ = attention_weights.permute(0, 2, 1, 3, 4).reshape(n * m, 1, lq, nl * p)
    out = (sampled * w_).sum(-1).reshape(n, m * d, lq)
    return out.transpose(1, 2).contiguous()


_lib = None


def _get_lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, level_start, loc, attn):
    dt = np.float64 if value.dtype == torch.float64 else np.float32
    v = np.ascontiguousarray(value.detach().cpu().numpy().astype(dt))
    l = np.ascontiguousarray(loc.detach().cpu().numpy().astype(dt))
    a = np.ascontiguousarray(attn.detach().cpu().numpy().astype(dt))
    s = np.ascontiguousarray(np.asarray(shapes.cpu()).astype(np.int64))
    ls = np.ascontiguousarray(np.asarray(level_start.cpu()).astype(np.int64))
    N, S, M, D = v.shape
    _, Lq, _, L, P, _ = l.shape
    return dt, v, s, ls, l, a, (N, S, M, D, L, Lq, P)


def msda_c_forward(value, shapes, level_start, loc, attn):
    dt, v, s, ls, l, a, dims = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=dt)
    fn = getattr(_get_lib(), "msda_oracle_fwd_f64" if dt == np.float64 else "msda_oracle_fwd_f32")
    fn(_ptr(v), _ptr(s), _ptr(ls), _ptr(l), _ptr(a), N, S, M, D, L, Lq, P, _ptr(out))
    return torch.from_numpy(out)


def msda_c_backward(value, shapes, level_start, loc, attn, grad_out):
    dt, v, s, ls, l, a, dims = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    go = np.ascontiguousarray(grad_out.detach().cpu().numpy().astype(dt))
    gv, gl, ga = np.empty_like(v), np.empty_like(l), np.empty_like(a)
    fn = getattr(_get_lib(), "msda_oracle_bwd_f64" if dt == np.float64 else "msda_oracle_bwd_f32")
    fn(_ptr(v), _ptr(s), _ptr(ls), _ptr(l), _ptr(a), _ptr(go), N, S, M, D, L, Lq, P, _ptr(gv), _ptr(gl), _ptr(ga))
    return torch.from_numpy(gv), torch.from_numpy(gl), torch.from_numpy(ga)


class MSDAOracleFunction(torch.autograd.Function):
    """autograd wrapper over the C oracle, signature of MSDeformAttnFunction.apply (ms_deform_attn_func.py:21-39)."""

    @staticmethod
    def forward(ctx, value, shapes, level_start, loc, attn, im2col_step=64):
        ctx.save_for_backward(value, shapes, level_start, loc, attn)
        return msda_c_forward(value, shapes, level_start, loc, attn).to(value.dtype)

    @staticmethod
    def backward(ctx, go):
        value, shapes, level_start, loc, attn = ctx.saved_tensors
        gv, gl, ga = msda_c_backward(value, shapes, level_start, loc, attn, go.contiguous())
        return gv.to(value.dtype), None, None, gl.to(loc.dtype), ga.to(attn.dtype), None
