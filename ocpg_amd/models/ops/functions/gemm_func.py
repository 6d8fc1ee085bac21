"""Row-major GEMM helpers over ocpg_gemm (csrc/gemm.hip: hipBLASLt with a per-shape plan cache)."""
import ctypes

import torch

from ...._lib import check, lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


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


def gemm_tn_split(a, b, splits):
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
    return part.sum(0)
