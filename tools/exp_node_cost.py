"""Experiment: wall-clock cost of one kernel node in a replayed HIP graph (a chain of N tiny dependent kernels), and of the same chain
when every kernel has some real work (1 MB / 16 MB elementwise)."""
import torch
dev = torch.device("cuda:0")


def chain(n_nodes, numel):
    x = torch.zeros(numel, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            x.add_(1.0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n_nodes):
            x.add_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3 / n_nodes      # us per node


for numel in (1, 1 << 18, 1 << 22, 1 << 24):
    for n in (200, 2000):
        print(f"{n:5d} nodes x {numel * 4 / 1e6:8.3f} MB add_: {chain(n, numel):7.2f} us per node", flush=True)
