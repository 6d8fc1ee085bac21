"""Diagnostic (r4): csrc/conv3x3_wgrad.hip at the ResNet-101 shapes of config #2 (10 frames) against im2col + row-split GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd._lib import lib, check
dev = torch.device("cuda:0")
L = lib()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, (n, c, co, h, w, s) in {"layer2 (128, 48x80)": (10, 128, 128, 48, 80, 1), "layer3 (256, 24x40)": (10, 256, 256, 24, 40, 1),
                                  "layer4 (512, 12x20)": (10, 512, 512, 12, 20, 1), "layer3.0 stride 2": (10, 256, 256, 48, 80, 2)}.items():
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    gz = torch.randn(n, ho, wo, co, device=dev).to(torch.bfloat16)
    sp = int(L.ocpg_conv3x3_mfma_wgrad_splits(n, h, w, c, co, s))
    part = torch.empty(sp, co, 9 * c, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: check(L.ocpg_conv3x3_mfma_wgrad(gz.data_ptr(), x.data_ptr(), n, h, w, c, co, s, part.data_ptr(), st), "wgrad"))
    fl = 2.0 * n * ho * wo * co * 9 * c
    print("%-22s splits %2d: %6.1f us = %5.0f TFLOP/s (%.3f of 2.5 PF)" % (name, sp, t, fl / t / 1e6, fl / t / 1e6 / 2500))
