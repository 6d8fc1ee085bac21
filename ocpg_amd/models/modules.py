"""LFMResizeAdaptive -- the spectral (low-frequency modulation) block used before and after text fusion.

Follows the reference's models/modules.py:9-61: a 3x3 *valid* "laplace" conv -> global average -> FC -> sigmoid
gives one coefficient per sample; the 2-D spectrum of the map is attenuated by (1 - coef * gaussian) where the
Gaussian (sigma 7) is centred at (h//2, w//2) of the UNSHIFTED spectrum and, once created at the first level, is
bilinearly resized for every later call; two 1x1 convs act on [real || imag]; inverse FFT; residual.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..util.misc import memo
from . import amp_cache
import os

from .ops.functions.spectral_func import (conv3x3_valid_spatial_mean, dft_supported, ifft2_real, lfm_inverse, lfm_spectrum, pair_to_complex,
                                          spectral_gate, spectral_gate_cl)

FUSED_GATE = True       # A/B switch: fused spectral gate kernel
GATE_NHWC = True        # A/B switch: gate output / inverse-FFT input in channels-last memory (1x1 convs as GEMMs, no casts / layout copies)
DFT_CL = os.environ.get("OCPG_LFM_DFT", "1") != "0"       # A/B switch: both transforms by csrc/lfm_dft.hip on the channels-last map (else rocFFT between transposes)
LAPLACE_MEAN = True     # A/B switch: mean(laplace(x)) as nine window means + one small matrix product (no convolution)


class LFMResizeAdaptive(nn.Module):
    def __init__(self, num_channels, sigma):
        super().__init__()
        self.conv1 = amp_cache.Conv2d(2 * num_channels, 2 * num_channels, kernel_size=1)
        self.conv2 = amp_cache.Conv2d(2 * num_channels, 2 * num_channels, kernel_size=1)
        amp_cache.mark_single_use(self.conv1, self.conv2)       # applied once per forward: their gradient sums ride in the fused cast
        self.sigma = sigma
        self.laplace = amp_cache.Conv2d(num_channels, num_channels, kernel_size=3, padding=0)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(amp_cache.Linear(num_channels, num_channels, bias=False), nn.ReLU(inplace=True),
                                amp_cache.Linear(num_channels, 1, bias=False), nn.Sigmoid())

    @staticmethod
    def make_gaussian(cy, cx, height, width, sigma=7, device="cpu"):
        ys = torch.arange(height, dtype=torch.float32, device=device).view(-1, 1)
        xs = torch.arange(width, dtype=torch.float32, device=device).view(1, -1)
        return torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * sigma ** 2))[None, None]

    def forward(self, x, gauss_map=None):
        # Precision follows the reference exactly: the input and the FFTs are fp32 (`x.float()`, modules.py:35), while the
        # three convs / two FCs are ordinary autocast ops (fp16 in the reference's --amp runs, bf16 here) whose outputs
        # are cast back with `.float()` (modules.py:56).  Without autocast everything is fp32.
        b, c, h, w = x.shape
        x = x.float()
        if x.is_cuda and LAPLACE_MEAN and self.laplace.kernel_size == (3, 3) and self.laplace.padding == (0, 0):
            # the spatial mean of a convolution is linear in its input (csrc/lfm.hip): same products, no 3x3 convolution
            amp = torch.is_autocast_enabled("cuda")
            wl, bl = amp_cache.lookup(self.laplace.weight), amp_cache.lookup(self.laplace.bias)
            if amp:
                dt = torch.get_autocast_dtype("cuda")
                wl, bl = wl.to(dt), bl.to(dt)
            lap_mean = conv3x3_valid_spatial_mean(x, wl, bl, amp and wl.dtype == torch.bfloat16)
            if amp:
                lap_mean = lap_mean.to(wl.dtype)          # the dtype the autocast convolution + mean would have produced
        else:
            lap_mean = self.laplace(x).mean(dim=(2, 3))
        coef = self.fc(lap_mean).view(b, 1, 1, 1)
        # the Gaussian of level 0 and its chain of bilinear resizes depend only on the map sizes: memoised (util.misc.memo)
        if gauss_map is None:
            key = ("gauss", h // 2, w // 2, h, w, float(self.sigma))
            high = memo("lfm_gauss", key, x.device, lambda: self.make_gaussian(h // 2, w // 2, h, w, self.sigma, x.device))
        else:
            parent = getattr(gauss_map, "_ocpg_key", None)
            key = None if parent is None else ("resized", parent, h, w)
            high = memo("lfm_gauss", key, x.device, lambda: F.interpolate(gauss_map, size=(h, w), mode="bilinear", align_corners=False))
        if key is not None:
            high._ocpg_key = key
        if x.is_cuda and FUSED_GATE and GATE_NHWC and c % 4 == 0:
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
            if DFT_CL and dft_supported(h, w):
                z = lfm_spectrum(x, coef.float().reshape(b), high.reshape(h, w), dt)      # fft2 + gate + [Re || Im], channels-last
                y = self.conv2(F.relu(self.conv1(z)))
                return lfm_inverse(y, x), high                                          # x + ifft2(.).real
            z = spectral_gate_cl(torch.fft.fft2(x), coef.float().reshape(b), high.reshape(h, w), dt)
            y = self.conv2(F.relu(self.conv1(z)))                   # channels-last in, channels-last out: two GEMMs
            y = ifft2_real(pair_to_complex(y), h, w)              # = torch.fft.ifft2(..., s=(h, w)).real, real-to-complex backward
            return x + y, high
        if x.is_cuda and FUSED_GATE:
            # gate, real/imag split and concatenation in one pass (csrc/spectral.hip)
            z = spectral_gate(torch.fft.fft2(x), coef.float().reshape(b), high.reshape(h, w))
        else:
            spec = torch.fft.fft2(x) * (1 - coef.float() * high)
            z = torch.cat([spec.real, spec.imag], dim=1)
        y = self.conv2(F.relu(self.conv1(z))).float()
        yr, yi = torch.chunk(y, 2, dim=1)
        y = torch.fft.ifft2(torch.complex(yr, yi), s=(h, w)).real.float()
        return x + y, high
