"""Diagnostic (r4): the convolutions MIOpen still served after round 3, library vs the HIP path, at BASELINE config #2's shapes
(10 frames of 384 x 640): stem 7x7/2 + BN + ReLU + maxpool, layer1's 3x3 64->64, the stride-2 projection shortcuts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ocpg_amd._lib import lib, check

dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


N = 10
# ---- layer1 3x3
x = torch.randn(N, 64, 96, 160, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=CL)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
w2 = w.permute(0, 2, 3, 1).contiguous()


def mine():
    check(lib().ocpg_conv3x3_mfma_fwd(x.data_ptr(), w2.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, N, 96, 160, 64, 64, 1, y.data_ptr(), st), "conv")


ref = F.relu(F.conv2d(x.float(), w.float(), padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
mine()
print("layer1 3x3 64->64: max err %.3e (ref max %.2f)" % ((y.float() - ref).abs().max().item(), ref.abs().max().item()))
print("  MIOpen conv only %.1f us; conv3x3_mfma (+BN+ReLU) %.1f us" % (timeit(lambda: F.conv2d(x, w, padding=1)), timeit(mine)))
# ---- stem
img = torch.randn(N, 3, 384, 640, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
w7 = (torch.randn(64, 3, 7, 7, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=CL)
print("stem: MIOpen 7x7/2 %.1f us; + maxpool %.1f us" % (timeit(lambda: F.conv2d(img, w7, stride=2, padding=3)),
                                                        timeit(lambda: F.max_pool2d(F.conv2d(img, w7, stride=2, padding=3), 3, 2, 1))))
# ---- strided 1x1 (layer2.0 / 3.0 / 4.0 downsample)
for c, co, h, wd in ((256, 512, 96, 160), (512, 1024, 48, 80), (1024, 2048, 24, 40)):
    xx = torch.randn(N, c, h, wd, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
    ww = (torch.randn(co, c, 1, 1, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=CL)
    t0 = timeit(lambda: F.conv2d(xx, ww, stride=2))
    t1 = timeit(lambda: F.conv2d(xx[:, :, ::2, ::2].contiguous(memory_format=CL), ww))
    print("1x1/2 %d->%d at %dx%d: MIOpen %.1f us, subsample + conv %.1f us" % (c, co, h, wd, t0, t1))
