"""Diagnostic: ResNet-101 3x3 convs at the bench shape: MIOpen vs im2col+GEMM (GPU-busy us per kernel, fwd+bwd)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from ocpg_amd.models import amp_cache
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
N = 10
shapes = [(64, 96, 160, 1), (128, 96, 160, 2), (128, 48, 80, 1), (256, 48, 80, 2), (256, 24, 40, 1), (512, 24, 40, 2), (512, 12, 20, 1)]
def busy(fn, n=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n): fn()
        torch.cuda.synchronize()
    ks = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    return sum(e.device_time_total for e in ks) / n, sorted(((e.device_time_total / n, e.count // n, e.key[:70]) for e in ks), reverse=True)
for c, h, w, s in shapes:
    conv = amp_cache.Conv2d(c, c, 3, stride=s, padding=1, bias=False).to(dev, torch.bfloat16).to(memory_format=torch.channels_last)
    x = torch.randn(N, c, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(N, c, (h - 1) // s + 1, (w - 1) // s + 1, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def fb():
        y = conv(x); y.backward(go); x.grad = None; conv.weight.grad = None
    out = []
    for on in (False, True):
        amp_cache.GEMM_3X3 = on
        t, ks = busy(fb)
        out.append((t, ks))
    print("C=%d %dx%d s%d: miopen %.0f us | im2col+gemm %.0f us" % (c, h, w, s, out[0][0], out[1][0]))
    for t, n, k in out[1][1]:
        print("      %7.1f us x%d %s" % (t, n, k))
    for t, n, k in out[0][1][:4]:
        print("   [miopen] %7.1f us x%d %s" % (t, n, k))
