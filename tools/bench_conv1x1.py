"""Diagnostic: ResNet-101 1x1 convs at the bench shape (10 frames 384x640, bf16 channels-last): MIOpen conv2d vs the same
contraction as a hipBLASLt GEMM on the NHWC view, forward and forward+backward."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
N = 10
shapes = [(64, 64, 96, 160), (64, 256, 96, 160), (256, 64, 96, 160), (256, 128, 96, 160), (128, 512, 48, 80), (512, 128, 48, 80),
          (512, 256, 48, 80), (256, 1024, 24, 40), (1024, 256, 24, 40), (1024, 512, 24, 40), (512, 2048, 12, 20), (2048, 512, 12, 20)]
from torch.profiler import profile, ProfilerActivity
def timeit(fn, n=10):
    """GPU-busy time per call (sum of kernel durations from the profiler): the loop itself is host-bound for small GEMMs."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return sum(e.device_time_total for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA) / n
print("%-26s %10s %10s | %10s %10s   (us; fwd, fwd+bwd)" % ("cin->cout @HxW", "conv fwd", "gemm fwd", "conv f+b", "gemm f+b"))
for cin, cout, h, w in shapes:
    x = torch.randn(N, cin, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(N, cout, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def conv_f():
        with torch.no_grad(): return F.conv2d(x, wt)
    def gemm_f():
        with torch.no_grad(): return F.linear(x.permute(0, 2, 3, 1), wt.view(cout, cin)).permute(0, 3, 1, 2)
    def conv_fb():
        y = F.conv2d(x, wt); y.backward(go); x.grad = None; wt.grad = None
    def gemm_fb():
        y = F.linear(x.permute(0, 2, 3, 1), wt.view(cout, cin)).permute(0, 3, 1, 2); y.backward(go); x.grad = None; wt.grad = None
    y1, y2 = conv_f(), gemm_f()
    assert y2.is_contiguous(memory_format=torch.channels_last) and (y1.float() - y2.float()).abs().max() <= 0.02 * y1.float().abs().max()
    gf = 2 * N * h * w * cin * cout / 1e6
    t = [timeit(f) for f in (conv_f, gemm_f, conv_fb, gemm_fb)]
    print("%4d->%4d @%3dx%3d %6.1fGF %10.1f %10.1f | %10.1f %10.1f" % (cin, cout, h, w, gf / 1e3, *t))
