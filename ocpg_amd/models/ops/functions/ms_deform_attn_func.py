"""MSDeformAttnFunction -- autograd binding of the HIP op; mirrors the reference's
models/ops/functions/ms_deform_attn_func.py:21-39 (same ``apply`` signature, same saved tensors, same
returned gradient tuple) but calls libocpg_hip.so through its C ABI instead of the pybind CUDA module.

Error behaviour follows ms_deform_attn_cuda.cu:28-38 / ms_deform_attn.h:38,60: non-contiguous or CPU tensors
raise RuntimeError.  ``im2col_step`` is accepted and ignored (no chunking, hence no ``N % step`` restriction).
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib, require_gpu, stream_ptr


# -- optional live kernel timing (bench.py): events on the launch stream around every kernel call -----------------
_TIMING = {"on": False, "events": []}


def enable_kernel_timing(on=True):
    _TIMING["on"] = on
    _TIMING["events"] = []


def collect_kernel_timing():
    """-> {"fwd_enc": {"ms": total, "n": launches}, ...}; 'enc' = self-attention shape (Lq == S), else 'dec'."""
    torch.cuda.synchronize()
    out = {}
    for key, e0, e1 in _TIMING["events"]:
        d = out.setdefault(key, {"ms": 0.0, "n": 0})
        d["ms"] += e0.elapsed_time(e1)
        d["n"] += 1
    _TIMING["events"] = []
    _TIMING["on"] = False
    return out


class _timed:
    def __init__(self, key):
        self.key = key

    def __enter__(self):
        if _TIMING["on"]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _TIMING["on"]:
            self.e1.record()
            _TIMING["events"].append((self.key, self.e0, self.e1))


def _host_shapes(spatial_shapes):
    """Host copy of the [L,2] shapes tensor. Our own transformer attaches it (no sync); foreign callers pay one
    D2H copy -- the reference's module syncs on the same tensor anyway (ms_deform_attn.py:94 assert)."""
    hs = getattr(spatial_shapes, "_ocpg_host", None)
    if hs is None:
        hs = spatial_shapes.detach().cpu().contiguous()
    return hs


def _dims(value, loc):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    return N, S, M, D, L, Lq, P


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    for n, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                 ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        require_gpu(n, t)
    if value.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("ms_deform_attn_forward: only float32 / float64 are supported")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("ms_deform_attn_forward: value / sampling_loc / attn_weight dtypes differ")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("ms_deform_attn_forward: spatial_shapes / level_start_index must be int64")
    N, S, M, D, L, Lq, P = _dims(value, sampling_loc)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device), _timed("fwd_enc" if Lq == S else "fwd_dec"):
        if value.dtype == torch.float32:
            # only the opt-in LDS-window forward reads it: never pay a device-to-host copy for it
            hs = getattr(spatial_shapes, "_ocpg_host", None) if Lq == S else None
            check(lib().ocpg_msda_fwd_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                          sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(),
                                          ctypes.c_void_p(hs.data_ptr()) if hs is not None else None, stream_ptr()),
                  "ocpg_msda_fwd")
        else:
            check(lib().ocpg_msda_fwd_f64(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                          sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(),
                                          stream_ptr()), "ocpg_msda_fwd")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step=64, sel_state=None):
    """sel_state: the call site's path-selection state (int32[8] on the device, zero-filled once; include/ocpg_hip.h
    ocpg_msda_bwd_value_sel_f32) or None for the fixed default path."""
    for n, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                 ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)):
        require_gpu(n, t)
    N, S, M, D, L, Lq, P = _dims(value, sampling_loc)
    grad_value = torch.zeros_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    if value.dtype == torch.float32 and Lq == S:
        # self-attention: the two halves of the backward are separate kernels behind their own entry points (timed apart)
        hs = _host_shapes(spatial_shapes)
        L_ = lib()
        with torch.cuda.device(value.device):
            with _timed("bwd_enc_value"):
                if sel_state is not None:
                    rc1 = L_.ocpg_msda_bwd_value_sel_f32(sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(), N, S, M, D, L,
                                                         Lq, P, grad_value.data_ptr(), ctypes.c_void_p(hs.data_ptr()), sel_state.data_ptr(),
                                                         stream_ptr())
                else:
                    rc1 = L_.ocpg_msda_bwd_value_f32(sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(), N, S, M, D, L,
                                                     Lq, P, grad_value.data_ptr(), ctypes.c_void_p(hs.data_ptr()), stream_ptr())
            if rc1 == 0:
                with _timed("bwd_enc_locattn"):
                    rc2 = L_.ocpg_msda_bwd_locattn_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                                       sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(), N, S,
                                                       M, D, L, Lq, P, grad_loc.data_ptr(), grad_attn.data_ptr(), stream_ptr())
                check(rc2, "ocpg_msda_bwd_locattn")
                return grad_value, grad_loc, grad_attn
            if rc1 != -2000:
                check(rc1, "ocpg_msda_bwd_value")
    with torch.cuda.device(value.device), _timed("bwd_enc" if Lq == S else "bwd_dec"):
        if value.dtype == torch.float32:
            hs = _host_shapes(spatial_shapes) if Lq == S else None
            check(lib().ocpg_msda_bwd_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                          sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                                          N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(),
                                          grad_attn.data_ptr(), ctypes.c_void_p(hs.data_ptr()) if hs is not None else None,
                                          stream_ptr()), "ocpg_msda_bwd")
        elif value.dtype == torch.float64:
            check(lib().ocpg_msda_bwd_f64(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                          sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                                          N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(),
                                          grad_attn.data_ptr(), stream_ptr()), "ocpg_msda_bwd")
        else:
            raise RuntimeError("ms_deform_attn_backward: only float32 / float64 are supported")
    return grad_value, grad_loc, grad_attn


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        ctx.shapes_host = getattr(value_spatial_shapes, "_ocpg_host", None)
        ctx.sel_state = getattr(sampling_locations, "_ocpg_sel", None)       # the calling module's path-selection state (or None)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, loc, attn = ctx.saved_tensors
        if ctx.shapes_host is not None:
            shapes._ocpg_host = ctx.shapes_host
        gv, gl, ga = ms_deform_attn_backward(value, shapes, level_start, loc, attn, grad_output.contiguous(), ctx.im2col_step, ctx.sel_state)
        return gv, None, None, gl, ga, None


class MSDeformAttnFusedFunction(Function):
    """The module's front end fused into the op (include/ocpg_hip.h: ocpg_msda_fused_fwd_f32 / _bwd_qproj_f32): softmax over the L*P logits
    and `reference + offset` (ms_deform_attn.py:96-110, 2-d reference branch) inside the forward kernel's sample setup, the softmax backward
    and the [d offsets | d logits] layout inside the gather kernel's epilogue.  Self-attention calls only (Lq == S), D = 32, L*P = 16,
    reference points without gradient; `supported()` says whether a call qualifies -- the module keeps the unfused path otherwise.

    apply(value [N,S,M,D], shapes, level_start, qproj [N,Lq,3*M*L*P], ref [N,Lq,L,2], L, P, sel_state) -> (out, loc, attn)"""

    @staticmethod
    def supported(value, qproj, ref, L, P):
        return (value.is_cuda and value.dtype == torch.float32 and qproj.dtype == torch.float32 and ref.dtype == torch.float32
                and value.shape[-1] == 32 and L * P == 16 and ref.shape[-1] == 2 and not ref.requires_grad
                and value.shape[1] == qproj.shape[1] and qproj.is_contiguous())

    @staticmethod
    def forward(ctx, value, shapes, level_start, qproj, ref, L, P, sel_state):
        N, S, M, D = value.shape
        Lq = qproj.shape[1]
        value, ref = value.contiguous(), ref.contiguous()
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        loc = torch.empty((N, Lq, M, L, P, 2), dtype=value.dtype, device=value.device)
        attn = torch.empty((N, Lq, M, L, P), dtype=value.dtype, device=value.device)
        with torch.cuda.device(value.device), _timed("fwd_enc"):
            check(lib().ocpg_msda_fused_fwd_f32(value.data_ptr(), shapes.data_ptr(), level_start.data_ptr(), qproj.data_ptr(), ref.data_ptr(),
                                                N, S, M, D, L, Lq, P, out.data_ptr(), loc.data_ptr(), attn.data_ptr(), stream_ptr()),
                  "ocpg_msda_fused_fwd")
        ctx.save_for_backward(value, shapes, level_start, loc, attn)
        ctx.shapes_host = getattr(shapes, "_ocpg_host", None)
        ctx.sel_state = sel_state
        ctx.qshape = qproj.shape
        ctx.mark_non_differentiable(loc, attn)
        return out, loc, attn

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, _gloc, _gattn):
        value, shapes, level_start, loc, attn = ctx.saved_tensors
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = loc.shape
        go = grad_output.contiguous()
        hs = ctx.shapes_host if ctx.shapes_host is not None else _host_shapes(shapes)
        grad_value = torch.zeros_like(value)
        grad_q = torch.empty(ctx.qshape, dtype=value.dtype, device=value.device)
        L_ = lib()
        with torch.cuda.device(value.device):
            with _timed("bwd_enc_value"):
                args = (loc.data_ptr(), attn.data_ptr(), go.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(), ctypes.c_void_p(hs.data_ptr()))
                if ctx.sel_state is not None:
                    rc = L_.ocpg_msda_bwd_value_sel_f32(*args, ctx.sel_state.data_ptr(), stream_ptr())
                else:
                    rc = L_.ocpg_msda_bwd_value_f32(*args, stream_ptr())
            if rc == -2000:        # shape not served by the self-attention scatter kernels: the whole backward through the generic entry point
                grad_value.zero_()
                gl, ga = torch.empty_like(loc), torch.empty_like(attn)
                check(L_.ocpg_msda_bwd_f32(value.data_ptr(), shapes.data_ptr(), level_start.data_ptr(), loc.data_ptr(), attn.data_ptr(), go.data_ptr(),
                                           N, S, M, D, L, Lq, P, grad_value.data_ptr(), gl.data_ptr(), ga.data_ptr(), ctypes.c_void_p(hs.data_ptr()),
                                           stream_ptr()), "ocpg_msda_bwd")
                glogit = attn.view(N, Lq, M, L * P) * (ga.view(N, Lq, M, L * P) - (attn * ga).view(N, Lq, M, L * P).sum(-1, keepdim=True))
                grad_q = torch.cat([gl.reshape(N, Lq, -1), glogit.reshape(N, Lq, -1)], -1)
                return grad_value, None, None, grad_q, None, None, None, None
            check(rc, "ocpg_msda_bwd_value")
            with _timed("bwd_enc_locattn"):
                check(L_.ocpg_msda_fused_bwd_qproj_f32(value.data_ptr(), shapes.data_ptr(), level_start.data_ptr(), loc.data_ptr(), attn.data_ptr(),
                                                       go.data_ptr(), N, S, M, D, L, Lq, P, grad_q.data_ptr(), stream_ptr()), "ocpg_msda_fused_bwd_qproj")
        return grad_value, None, None, grad_q, None, None, None, None
