"""MSDeformAttn module -- multi-scale deformable attention around the HIP op.

Same parameters, initialisation and forward contract as the reference's
models/ops/modules/ms_deform_attn.py:31-118 (returns (output, sampling_locations, attention_weights)).
Runs in fp32 regardless of autocast, as the reference does (deformable_transformer.py:250,329).
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from ...amp_cache import TokenLinear
from ..functions import MSDeformAttnFunction


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("d_model // n_heads should be a power of 2 (the fast HIP kernel needs it; others take the generic kernel)")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = TokenLinear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = TokenLinear(d_model, n_heads * n_levels * n_points)
        self.value_proj = TokenLinear(d_model, d_model)
        self.output_proj = TokenLinear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """Zero offset/attention weights; offset bias = 8-direction ring scaled by the point index (ms_deform_attn.py:64-78)."""
        nn.init.zeros_(self.sampling_offsets.weight)
        theta = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack([theta.cos(), theta.sin()], -1)
        ring = ring / ring.abs().max(-1, keepdim=True)[0]
        ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(ring.reshape(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        N, Lq, _ = query.shape
        _, S, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        host = getattr(input_spatial_shapes, "_ocpg_host", None)
        if host is not None:
            assert int((host[:, 0] * host[:, 1]).sum()) == S
        else:
            assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == S

        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, M, self.d_model // M)
        offsets = self.sampling_offsets(query).view(N, Lq, M, L, P, 2)
        weights = F.softmax(self.attention_weights(query).view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        if reference_points.shape[-1] == 2:
            wh = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            loc = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.contiguous(), weights.contiguous(), self.im2col_step)
        return self.output_proj(out), loc, weights
