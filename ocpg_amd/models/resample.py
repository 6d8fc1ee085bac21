"""Separable resampling as two small matrix products.

`F.interpolate(x, size, mode="bicubic", align_corners=False)` (ocpg.py:337) is linear and separable:
out = Wy @ x @ Wx^T per channel, with the PyTorch bicubic taps (A = -0.75, source index (dst + 0.5) * in/out - 0.5,
neighbour indices clamped to the map).  Written this way the BACKWARD is two more matrix products instead of
ATen's atomics-based upsample_bicubic2d_backward kernel, which alone cost 40 ms per call at 10 x 256 x 48 x 80 on
MI355X (profiles/steady_r1a.csv) -- 48 % of the whole training step.
"""
import torch

_CACHE = {}


def _cubic_taps(t, a=-0.75):
    def w1(x):   # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1

    def w2(x):   # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    return torch.stack([w2(t + 1), w1(t), w1(1 - t), w2(2 - t)], -1)


def bicubic_matrix(n_in, n_out, device):
    """[n_out, n_in] fp32 matrix of 1-D bicubic resampling (align_corners=False), built once per (sizes, device)."""
    key = (n_in, n_out, str(device))
    if key not in _CACHE:
        dst = torch.arange(n_out, dtype=torch.float32)
        src = (dst + 0.5) * (n_in / n_out) - 0.5
        i0 = torch.floor(src)
        taps = _cubic_taps(src - i0)                                               # [n_out, 4]
        idx = (i0.long()[:, None] + torch.arange(-1, 3)[None, :]).clamp(0, n_in - 1)  # border: clamp the neighbour index
        m = torch.zeros(n_out, n_in)
        m.scatter_add_(1, idx, taps)
        _CACHE[key] = m.to(device)
    return _CACHE[key]


def bicubic_resize(x, size):
    """x [N, C, h, w] -> [N, C, H, W]; identity when the size already matches (scale-1 bicubic is exact identity)."""
    h, w = x.shape[-2:]
    H, W = size
    if (h, w) == (H, W):
        return x
    wy = bicubic_matrix(h, H, x.device).to(x.dtype)
    wx = bicubic_matrix(w, W, x.device).to(x.dtype)
    return torch.matmul(wy, torch.matmul(x, wx.t()))
