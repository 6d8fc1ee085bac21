"""Plain multi-head attention with nn.MultiheadAttention's parameter names (in_proj_weight, in_proj_bias,
out_proj.{weight,bias}) so reference checkpoints load; sequence-first [L, B, C] like the reference's call sites
(deformable_transformer.py:325, segmentation.py:108-111)."""
import torch
import torch.nn.functional as F
from torch import nn

from . import amp_cache
from .amp_cache import lookup


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = amp_cache.Linear(embed_dim, embed_dim)
        amp_cache.register(self, self.in_proj_weight, self.in_proj_bias)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, query, key, value, key_padding_mask=None):
        """query [Lq,B,C], key/value [Lk,B,C], key_padding_mask [B,Lk] (True = ignore) -> [Lq,B,C]."""
        Lq, B, C = query.shape
        Lk = key.shape[0]
        H, hd = self.num_heads, C // self.num_heads
        w, b = lookup(self.in_proj_weight), lookup(self.in_proj_bias)
        q = F.linear(query, w[:C], b[:C]).view(Lq, B, H, hd).permute(1, 2, 0, 3)
        k = F.linear(key, w[C:2 * C], b[C:2 * C]).view(Lk, B, H, hd).permute(1, 2, 0, 3)
        v = F.linear(value, w[2 * C:], b[2 * C:]).view(Lk, B, H, hd).permute(1, 2, 0, 3)
        mask = None
        if key_padding_mask is not None:
            mask = torch.zeros((B, 1, 1, Lk), dtype=q.dtype, device=q.device).masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=self.dropout if self.training else 0.0)
        return self.out_proj(o.permute(2, 0, 1, 3).reshape(Lq, B, C))
