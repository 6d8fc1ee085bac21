"""Micro-benchmark: csrc/conv3x3_mfma.hip (fwd + BN/ReLU epilogue; bwd = bn_act_bwd + MFMA dgrad + im2col/GEMM wgrad) against
MIOpen's conv + the frozen-BN kernel, at the ResNet-101 3x3 shapes of BASELINE config #2 (10 frames, bf16, channels-last)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd.models import amp_cache, backbone
from ocpg_amd.models.ops.functions import conv_bn_func as f

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (n, c, h, w, s) in ((10, 128, 48, 80, 1), (10, 256, 24, 40, 1), (10, 512, 12, 20, 1), (10, 256, 48, 80, 2), (10, 512, 24, 40, 2)):
    x = torch.randn(n, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(c, c, 3, 3, device=dev) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    go = torch.randn(n, c, ho, wo, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * n * ho * wo * c * c * 9

    def mfma_f():
        return f.conv3x3_mfma_bn_act(x, wt, scale, shift, True, s, 1)

    def mfma_fb():
        torch.autograd.grad(mfma_f(), (x, wt), go)

    def mio_f():
        y = torch.nn.functional.conv2d(x, wt, None, s, 1)
        return backbone.bn_act_func.frozen_bn_act(y, scale, shift, None, True) if hasattr(backbone, "bn_act_func") else (y * scale.view(1, -1, 1, 1).to(y.dtype) + shift.view(1, -1, 1, 1).to(y.dtype)).relu()

    def mio_fb():
        torch.autograd.grad(mio_f(), (x, wt), go)
    with torch.no_grad():
        tf_m, tf_o = timeit(mfma_f), timeit(mio_f)
    tb_m, tb_o = timeit(mfma_fb), timeit(mio_fb)
    print(f"{c:4d}ch {h}x{w}/s{s}: fwd mfma {tf_m:7.1f} us ({flops / tf_m / 1e6:6.1f} TFLOP/s)  miopen+bn {tf_o:7.1f} us | "
          f"fwd+bwd mfma {tb_m:7.1f} us  miopen+bn {tb_o:7.1f} us", flush=True)
