"""Diagnostic: are consecutive kernel nodes of a captured chain really ordered on replay?  A chain of (zero-fill -> atomic
index_add -> read) triples on many buffers; every final value is known."""
import os, sys, torch
dev = torch.device("cuda:0")
n, pairs = 1 << 16, int(os.environ.get("PAIRS", "400"))
idx = torch.randint(0, n, (1 << 20,), device=dev)
ones = torch.ones(1 << 20, device=dev)
want = torch.zeros(n, device=dev).index_add_(0, idx, ones)
bufs = [torch.full((n,), 5.0, device=dev) for _ in range(8)]
outs = []
def build():
    outs.clear()
    for i in range(pairs):
        b = bufs[i % 8]
        b.zero_()
        b.index_add_(0, idx, ones)
        outs.append((b * 1.0).sum())
    return outs
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    build()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g, stream=side):
    res = list(build())
if os.environ.get("REPAIR", "1") == "1":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ctypes
    from ocpg_amd import _lib
    k = ctypes.c_int(0)
    _lib.check(_lib.lib().ocpg_graph_replace_memsets(g.raw_cuda_graph(), ctypes.byref(k)), "x")
    print("memset nodes replaced:", k.value)
g.instantiate()
total = float(want.sum())
for r in range(5):
    g.replay(); torch.cuda.synchronize()
    vals = torch.stack(res).tolist()
    bad = [(i, v) for i, v in enumerate(vals) if abs(v - total) > 1e-3 * total]
    print("replay", r, "wrong sums:", len(bad), "of", pairs, bad[:4], "expected", total, flush=True)
