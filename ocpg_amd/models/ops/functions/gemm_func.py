"""Row-major GEMM helpers over ocpg_gemm (csrc/gemm.hip: hipBLASLt with a per-shape plan cache)."""
import ctypes

import torch

from ...._lib import check, lib

import os

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
# A/B switch: the path's large GEMMs (token Linears, 1x1 / im2col convolutions, their gradients) through the plan cache, whose plans
# are timed at first use (csrc/gemm.hip: tune()); "0" = torch.mm / addmm / bmm (hipBLASLt's first heuristic choice)
PLANNED = os.environ.get("OCPG_PLANNED_GEMM", "1") != "0"


def _plain(t):
    return t.dim() == 2 and t.is_cuda and t.dtype in _DT and t.shape[0] > 0 and t.shape[1] > 0 and (t.stride(1) == 1 or t.shape[1] == 1) \
        and (t.shape[0] == 1 or t.stride(0) >= t.shape[1])


def mm(a, b, trans_b=False, bias=None):
    """a [M,K] @ (b [K,N] or, with trans_b, b [N,K]^T) (+ bias [N]): torch.mm / addmm semantics for 2-D GPU operands, through the
    plan cache when the operands are plain row-major matrices of one dtype."""
    if PLANNED and _plain(a) and _plain(b) and a.dtype == b.dtype and (bias is None or (bias.dtype == a.dtype and bias.is_contiguous())):
        return gemm(a, b, False, trans_b, bias)
    bb = b.t() if trans_b else b
    return torch.mm(a, bb) if bias is None else torch.addmm(bias, a, bb)


def mm_tn(a, b, splits=1, defer=False):
    """a [R,M]^T @ b [R,N] -> [M,N] (weight-gradient form), optionally reduced over `splits` row chunks (batched GEMM + sum; with
    `defer` the sum is left to the fused gradient cast: amp_cache.defer_sum)."""
    if PLANNED and _plain(a) and _plain(b) and a.dtype == b.dtype:
        if splits == 1:
            return gemm(a, b, True, False)
        if a.is_contiguous() and b.is_contiguous():
            return gemm_tn_split(a, b, splits, defer)
    if splits == 1:
        return torch.mm(a.t(), b)
    r = a.shape[0]
    return torch.bmm(a.view(splits, r // splits, -1).transpose(1, 2), b.view(splits, r // splits, -1)).sum(0)


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def gemm(a, b, trans_a=False, trans_b=False, bias=None):
    """op(a) [M,K] @ op(b) [K,N] (+ bias[N]) for 2-D GPU tensors with unit inner stride -> new [M,N] tensor."""
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    n = b.shape[0] if trans_b else b.shape[1]
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    dt = _DT[a.dtype]
    rc = lib().ocpg_gemm(a.data_ptr(), b.data_ptr(), out.data_ptr(), None if bias is None else bias.data_ptr(), dt, dt, trans_a, trans_b,
                         m, n, k, _ld(a), _ld(b), n, 1, 0, 0, 0, 1.0, 0.0, torch.cuda.current_stream().cuda_stream)
    if rc:
        check(rc, "ocpg_gemm")
    return out


def gemm_tn_split(a, b, splits, defer=False):
    """a [R, M], b [R, N] (row-major, R = splits * r) -> a^T b [M, N], reduced over `splits` row chunks as one strided-batched
    GEMM + one sum (see amp_cache.weight_grad for why)."""
    r = a.shape[0] // splits
    m, n = a.shape[1], b.shape[1]
    part = torch.empty((splits, m, n), dtype=a.dtype, device=a.device)
    dt = _DT[a.dtype]
    rc = lib().ocpg_gemm(a.data_ptr(), b.data_ptr(), part.data_ptr(), None, dt, dt, 1, 0, m, n, r, m, n, n, splits, r * m, r * n, m * n,
                         1.0, 0.0, torch.cuda.current_stream().cuda_stream)
    if rc:
        check(rc, "ocpg_gemm")
    if defer:
        from ...amp_cache import defer_sum
        return defer_sum(part)
    return part.sum(0)
