"""Sine positional encodings (reference models/position_encoding.py: 1-D for text :12-45, 2-D for maps :48-84)."""
import math

import torch
from torch import nn

from ..util.misc import mask_key, memo


_FREQ_CACHE = {}


def _freq(num_pos_feats, temperature, device):
    """temperature ** (2*(k//2)/feats) as a host-computed constant table.

    Computed with CPU libm and then copied to the device (cached): in padded rows/columns the normalised coordinate
    is ~ -0.5/1e-6*2pi = -3e6, where a 1-ulp difference between a device `pow` and the host `pow` flips the sine.
    With a common table the GPU encoding matches the (CPU-generated) reference vectors to fp32 rounding."""
    key = (num_pos_feats, float(temperature), str(device))
    if key not in _FREQ_CACHE:
        k = torch.arange(num_pos_feats, dtype=torch.float32)
        _FREQ_CACHE[key] = (temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / num_pos_feats)).to(device)
    return _FREQ_CACHE[key]


def _interleave_sin_cos(x):
    """[..., C] -> sin on even / cos on odd feature indices, interleaved back to [..., C]."""
    return torch.stack((x[..., 0::2].sin(), x[..., 1::2].cos()), dim=-1).flatten(-2)


class PositionEmbeddingSine1D(nn.Module):
    def __init__(self, num_pos_feats=256, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list):
        mask = tensor_list.mask                                           # [B, L]
        assert mask is not None
        pos = (~mask).cumsum(1, dtype=torch.float32)
        if self.normalize:
            pos = pos / (pos[:, -1:] + 1e-6) * self.scale
        freq = _freq(self.num_pos_feats, self.temperature, mask.device)
        return _interleave_sin_cos(pos[:, :, None] / freq).permute(0, 2, 1)   # [B, C, L]


class PositionEmbeddingSine2D(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list):
        mask = tensor_list.mask                                           # [B, H, W]
        assert mask is not None
        cfg = ("pos2d", self.num_pos_feats, float(self.temperature), self.normalize, float(self.scale))
        return memo(cfg, mask_key(mask), mask.device, lambda: self._encode(mask))

    def _encode(self, mask):
        valid = ~mask
        y = valid.cumsum(1, dtype=torch.float32)
        x = valid.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y = (y - 0.5) / (y[:, -1:, :] + 1e-6) * self.scale
            x = (x - 0.5) / (x[:, :, -1:] + 1e-6) * self.scale
        freq = _freq(self.num_pos_feats, self.temperature, mask.device)
        px = _interleave_sin_cos(x[..., None] / freq)
        py = _interleave_sin_cos(y[..., None] / freq)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)              # [B, 2*feats, H, W]


def build_position_encoding(args):
    if args.position_embedding not in ("v2", "sine"):
        raise ValueError(f"not supported {args.position_embedding}")
    return PositionEmbeddingSine2D(args.hidden_dim // 2, normalize=True)
