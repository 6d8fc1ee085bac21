"""Diagnostic (r4): the LFM transforms of csrc/lfm_dft.hip per pyramid level at BASELINE config #2's shapes (10 frames, 256 channels),
against torch.fft on planes.  Bytes = the passes' algorithmic traffic (real map + half spectrum, half spectrum + pair)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd.models.ops.functions import spectral_func as sf

dev = torch.device("cuda:0")


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


N, C = 10, 256
for h, w in ((48, 80), (24, 40), (12, 20), (6, 10)):
    x = torch.randn(N, h, w, C, device=dev)
    coef, high = torch.rand(N, device=dev), torch.rand(h, w, device=dev)
    pair = sf._spectrum(x, coef, high, 1.0, torch.bfloat16)
    t_f = timeit(lambda: sf._spectrum(x, coef, high, 1.0, torch.bfloat16))
    t_i = timeit(lambda: sf._inverse(pair, None, None, None, False, 1.0 / (h * w), x))
    t_b = timeit(lambda: sf._inverse(pair, coef, high, pair, True, 1.0, None))
    xp = x.permute(0, 3, 1, 2).contiguous()
    t_r = timeit(lambda: torch.fft.fft2(xp))
    real, half, pr = N * h * w * C * 4, N * h * (w // 2 + 1) * C * 8, N * h * w * 2 * C * 2
    print("%2dx%2d: spectrum (rows_fwd + cols_fwd) %6.1f us [%4.0f GB/s]  inverse+residual %6.1f us [%4.0f GB/s]  gated inverse + dcoef %6.1f us   (torch.fft.fft2 alone %6.1f us)"
          % (h, w, t_f, (real + 2 * half + pr) / t_f / 1e3, t_i, (pr + 2 * half + 2 * real) / t_i / 1e3, t_b, t_r))
