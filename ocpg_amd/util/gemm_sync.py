"""Rank-consistent hipBLASLt plan choices for data-parallel runs (include/ocpg_hip.h: ocpg_gemm_set_tuning / _export_picks / _import_picks).

The plan cache of csrc/gemm.hip times the heuristic's ranked candidates the first time a bf16 / fp16 GEMM shape is used and keeps the
fastest -- a per-process measurement.  With N ranks each measuring for itself, two ranks can settle on different kernels for the same
shape, and the slowest choice sets every step (the gradient all-reduce waits for it).  Protocol (what main.py:62's DDP gets for free
from every rank linking the same cuBLAS heuristics):
    every rank but `src` calls `follow()` before its first GEMM            (no candidate timing there)
    all ranks run the same warm-up step                                    (rank `src` times its plans meanwhile)
    all ranks call `share(group)`                                          (src's choices are broadcast and imported everywhere)
"""
import ctypes

import torch
import torch.distributed as dist


def follow():
    """This process does not time candidates: it will run what the tuning rank picked (until then the heuristic's first choice)."""
    from .._lib import lib
    lib().ocpg_gemm_set_tuning(0)


def export_picks():
    """-> int64 tensor [n, 2] of (plan key hash, candidate index) for every plan this process has timed."""
    from .._lib import lib
    n = int(lib().ocpg_gemm_export_picks(None, 0))
    buf = torch.zeros((max(n, 0), 2), dtype=torch.int64)
    if n > 0:
        got = int(lib().ocpg_gemm_export_picks(ctypes.c_void_p(buf.data_ptr()), n))
        buf = buf[:min(n, got)]
    return buf


def import_picks(picks):
    from .._lib import check, lib
    picks = picks.to("cpu", torch.int64).contiguous()
    check(lib().ocpg_gemm_import_picks(ctypes.c_void_p(picks.data_ptr()) if picks.numel() else None, picks.shape[0]), "ocpg_gemm_import_picks")


def share(device, src=0, group=None, export=export_picks, apply=import_picks):
    """Broadcast rank `src`'s plan choices and import them on every other rank.  Returns the number of choices shared.  (`export` /
    `apply` are injectable: the 2-rank gloo test drives the protocol without a GPU.)"""
    rank = dist.get_rank(group)
    picks = export() if rank == src else None
    n = torch.tensor([picks.shape[0] if picks is not None else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src, group=group)
    buf = picks.to(device) if picks is not None else torch.zeros((int(n.item()), 2), dtype=torch.int64, device=device)
    if int(n.item()):
        dist.broadcast(buf, src=src, group=group)
    if rank != src:
        apply(buf)
    return int(n.item())
