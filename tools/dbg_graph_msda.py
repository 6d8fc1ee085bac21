"""Diagnostic: the MSDeformAttn backward alone under HIP-graph replay (tiny decoder / encoder shapes and the bench shapes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ocpg_amd.models.ops.functions import ms_deform_attn_func as F
dev = torch.device("cuda:0")
def case(name, N, shapes_hw, M, D, Lq_mode, P):
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = torch.tensor(shapes_hw, dtype=torch.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum()); L = len(shapes_hw)
    Lq = S if Lq_mode == "self" else Lq_mode
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    sh_d, st_d = shapes.to(dev), start.to(dev)
    sh_d._ocpg_host = shapes; st_d._ocpg_host = start
    value = torch.randn(N, S, M, D, device=dev, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, device=dev, generator=g) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, device=dev, generator=g), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, device=dev, generator=g)
    want = F.ms_deform_attn_backward(value, sh_d, st_d, loc, attn, go)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        F.ms_deform_attn_backward(value, sh_d, st_d, loc, attn, go)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        # some unrelated allocations / frees around it, as in the step
        junk = torch.randn(1000, device=dev) * 2
        out = F.ms_deform_attn_backward(value, sh_d, st_d, loc, attn, go)
        del junk
    res = []
    for r in range(6):
        gr.replay(); torch.cuda.synchronize()
        res.append(tuple(round(((a - b).abs().max() / (b.abs().max() + 1e-20)).item(), 7) for a, b in zip(out, want)))
        for o in out: o.add_(1.0)        # dirty the static outputs between replays
    print(name, res, flush=True)
case("tiny decoder", 10, [(8, 8), (4, 4), (2, 2), (1, 1)], 8, 32, 5, 4)
case("tiny encoder", 10, [(8, 8), (4, 4), (2, 2), (1, 1)], 8, 32, "self", 4)
case("bench decoder", 10, [(48, 80), (24, 40), (12, 20), (6, 10)], 8, 32, 5, 4)
case("bench encoder", 10, [(48, 80), (24, 40), (12, 20), (6, 10)], 8, 32, "self", 4)
