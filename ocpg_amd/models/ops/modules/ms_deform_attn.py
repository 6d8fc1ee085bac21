"""MSDeformAttn module -- multi-scale deformable attention around the HIP op.

Same parameters, initialisation and forward contract as the reference's
models/ops/modules/ms_deform_attn.py:31-118 (returns (output, sampling_locations, attention_weights)).
Runs in fp32 regardless of autocast, as the reference does (deformable_transformer.py:250,329).
"""
import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from ...amp_cache import TokenLinear, linear
from ....util.misc import memo
from ..functions import MSDeformAttnFunction
from ..functions.ms_deform_attn_func import MSDeformAttnFusedFunction


FUSED_FRONT = os.environ.get("OCPG_MSDA_FUSED_FRONT", "1") != "0"   # A/B switch: softmax + location arithmetic (and their backward) inside the op's kernels
SELECT_PATH = os.environ.get("OCPG_MSDA_SELECT", "1") != "0"     # A/B switch: per-call choice of the grad_value kernel family (self-attention calls)
MERGED_QUERY_PROJ = True      # A/B switch: sampling_offsets and attention_weights as ONE GEMM over the query (they share their input)


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
        # path-selection state of this module's backward (include/ocpg_hip.h: ocpg_msda_bwd_value_sel_f32): which grad_value kernel family the
        # next call takes, kept on the device by the kernels themselves.  Not part of the state_dict (checkpoints stay interchangeable).
        self.register_buffer("_sel_state", torch.zeros(8, dtype=torch.int32), persistent=False)
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
        if MERGED_QUERY_PROJ and query.is_cuda:
            # One GEMM for both query projections (same input, [256 + 128] output columns): one forward GEMM, one input-gradient
            # GEMM and one weight-gradient GEMM instead of two each plus the add of the two input gradients.  With 2-d reference
            # points and host-known level shapes the division of the offsets by (W_l, H_l) is folded into the (tiny) weight:
            # x (W s)^T + b s == (x W^T + b) s column by column -- same products, the scaling moves from 52 MB of offsets to 256 rows.
            n_off = M * L * P * 2
            so_w, so_b = self.sampling_offsets.weight, self.sampling_offsets.bias
            folded = reference_points.shape[-1] == 2 and host is not None
            if folded:
                key = ("msda_inv_wh", tuple(host.flatten().tolist()), M, P)
                inv = memo("msda", key, query.device, lambda: (1.0 / torch.stack([host[:, 1], host[:, 0]], -1).float())[None, :, None, :]
                           .expand(M, L, P, 2).reshape(-1).to(query.device))
                so_w, so_b = so_w * inv[:, None], so_b * inv
            both = linear(query, torch.cat([so_w, self.attention_weights.weight], 0), torch.cat([so_b, self.attention_weights.bias], 0))
            if (FUSED_FRONT and folded and Lq == S and MSDeformAttnFusedFunction.supported(value, both, reference_points, L, P)):
                # lines 96-110 of the reference module inside the kernels: no softmax / add / split-cat passes over the [N, Lq, 384] projection
                out, loc, weights = MSDeformAttnFusedFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index, both,
                                                                    reference_points, L, P, self._sel_state if SELECT_PATH else None)
                return self.output_proj(out), loc, weights
            off2, logit2 = torch.split(both, [n_off, M * L * P], dim=-1)      # split: its backward is ONE cat (two slices: 2 x (zeros + copy) + add)
            offsets = off2.view(N, Lq, M, L, P, 2)
            weights = F.softmax(logit2.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        else:
            folded = False
            offsets = self.sampling_offsets(query).view(N, Lq, M, L, P, 2)
            weights = F.softmax(self.attention_weights(query).view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        if reference_points.shape[-1] == 2:
            if folded:
                loc = reference_points[:, :, None, :, None, :] + offsets
            else:
                wh = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
                loc = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        loc_c = loc.contiguous()
        if SELECT_PATH and Lq == S and loc_c.is_cuda:
            loc_c._ocpg_sel = self._sel_state
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc_c, weights.contiguous(), self.im2col_step)
        return self.output_proj(out), loc, weights
