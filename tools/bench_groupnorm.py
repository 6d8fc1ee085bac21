"""Timing of csrc/groupnorm.hip at the input-projection shapes of config #2 (10 frames, bf16 maps) against autocast's ATen path
(cast + contiguous + native_group_norm and its backward).  GPU-busy time per call from HIP events over 50 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd.models.ops.functions.groupnorm_func import GroupNorm

dev = torch.device("cuda:0")
def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (n, c, h, w) in ((10, 256, 48, 80), (10, 256, 24, 40), (10, 256, 12, 20), (10, 256, 6, 10)):
    gn = GroupNorm(32, c).to(dev)
    x = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(n, c, h, w, device=dev)
    for name, f in (("hip ", lambda: gn(x)), ("aten", lambda: torch.nn.functional.group_norm(x.float(), 32, gn.weight, gn.bias, gn.eps))):
        y = f()
        tf = timed(f)
        tb = timed(lambda: torch.autograd.grad(y, [x, gn.weight, gn.bias], go, retain_graph=True))
        mb = n * c * h * w * 6 / 1e6
        print("%s N=%d %dx%d  fwd %7.1f us (%.2f TB/s)  bwd %7.1f us (%.2f TB/s)" % (name, n, h, w, tf, mb / tf, tb, n * c * h * w * 8 / 1e6 / tb))
