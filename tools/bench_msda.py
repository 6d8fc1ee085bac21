"""Micro-benchmark of the MSDeformAttn kernels at BASELINE config #2 encoder/decoder shapes (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd.models.ops.functions import ms_deform_attn_forward, ms_deform_attn_backward

dev = torch.device("cuda:0")
shapes_l = [(48, 80), (24, 40), (12, 20), (6, 10)]
shapes = torch.tensor(shapes_l, dtype=torch.long)
ls = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S = int(shapes.prod(1).sum())
N, M, D, L, P = int(os.environ.get("MSDA_FRAMES", "5")), 8, 32, 4, 4


def inputs(Lq, mode):
    g = torch.Generator().manual_seed(0)
    value = torch.randn(N, S, M, D, generator=g)
    if mode == "ring" and Lq == S:   # the reference's init: reference point = own pixel, ring offsets of 1..4 px
        refs = []
        for (h, w) in shapes_l:
            ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
            refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
        ref = torch.cat(refs, 0)[None, :, None, None, None, :]
        import math
        th = torch.arange(M) * (2 * math.pi / M)
        grid = torch.stack([th.cos(), th.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True)[0]
        off = grid.view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1).view(1, 1, 1, 1, P, 1)
        norm = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        loc = (ref + off / norm).expand(N, S, M, L, P, 2).contiguous()
    else:
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    return [t.to(dev) for t in (value, shapes, ls, loc, attn)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


for Lq, mode in ((S, "ring"), (S, "uniform"), (5, "uniform")):
    value, sh, lsi, loc, attn = inputs(Lq, mode)
    sh._ocpg_host = shapes
    go = torch.randn(N, Lq, M * D, device=dev)
    fwd_bytes = 4 * N * (S * M * D + Lq * M * D + 3 * Lq * M * L * P)
    bwd_bytes = fwd_bytes + 4 * N * (Lq * M * D) + 4 * N * (S * M * D + 3 * Lq * M * L * P)
    t_f = timeit(lambda: ms_deform_attn_forward(value, sh, lsi, loc, attn))
    t_b = timeit(lambda: ms_deform_attn_backward(value, sh, lsi, loc, attn, go))
    print(f"Lq={Lq:5d} {mode:8s} fwd {t_f:8.1f} us ({fwd_bytes / t_f / 1e6:6.2f} TB/s algo)   "
          f"bwd(+zero fill) {t_b:8.1f} us ({bwd_bytes / t_b / 1e6:6.2f} TB/s algo)", flush=True)
