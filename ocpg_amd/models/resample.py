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


def bilinear_matrix(n_in, n_out, align_corners, device):
    """[n_out, n_in] fp32 matrix of 1-D bilinear resampling with ATen's source-index rules (upsample_bilinear2d: align_corners=True
    -> dst * (in - 1) / (out - 1); False -> max((dst + 0.5) * in / out - 0.5, 0); second tap clamped to the last index)."""
    key = ("bilinear", n_in, n_out, bool(align_corners), str(device))
    if key not in _CACHE:
        dst = torch.arange(n_out, dtype=torch.float32)
        if align_corners:
            src = dst * ((n_in - 1) / (n_out - 1)) if n_out > 1 else torch.zeros_like(dst)
        else:
            src = ((dst + 0.5) * (n_in / n_out) - 0.5).clamp(min=0)
        i0 = torch.floor(src).clamp(max=n_in - 1)
        lam = src - i0
        i0 = i0.long()
        i1 = (i0 + 1).clamp(max=n_in - 1)
        m = torch.zeros(n_out, n_in)
        m.scatter_add_(1, i0[:, None], (1 - lam)[:, None])
        m.scatter_add_(1, i1[:, None], lam[:, None])
        _CACHE[key] = m.to(device)
    return _CACHE[key]


def bilinear_resize(x, size, align_corners):
    """F.interpolate(x, size, mode="bilinear", align_corners=...) for [N, C, h, w] as two matrix products in fp32 (ATen's backward,
    upsample_bilinear2d_backward, scatters with float atomics: 175 us per call at the 192 x 320 level-set maps); the result is
    returned in x's dtype, as F.interpolate does."""
    h, w = x.shape[-2:]
    H, W = size
    wy = bilinear_matrix(h, H, align_corners, x.device)
    wx = bilinear_matrix(w, W, align_corners, x.device)
    with torch.autocast(device_type=x.device.type, enabled=False):
        y = torch.matmul(wy, torch.matmul(x.float(), wx.t()))
    return y.to(x.dtype)


class _NearestUp(torch.autograd.Function):
    """F.interpolate(x, scale_factor=k) (nearest, integer k): every pixel becomes a k x k block, so the backward is a k x k block sum
    (one pooling kernel; ATen's upsample_nearest2d_backward took 120 us on the [40, 1, 384, 640] refined masks)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        return torch.nn.functional.interpolate(x, scale_factor=k)

    @staticmethod
    def backward(ctx, g):
        k = ctx.k
        return torch.nn.functional.avg_pool2d(g, k) * float(k * k), None


def nearest_upsample(x, k):
    return _NearestUp.apply(x, int(k))
