"""Plain multi-head attention with nn.MultiheadAttention's parameter names (in_proj_weight, in_proj_bias,
out_proj.{weight,bias}) so reference checkpoints load; sequence-first [L, B, C] like the reference's call sites
(deformable_transformer.py:325, segmentation.py:108-111)."""
import torch
import torch.nn.functional as F
from torch import nn

from . import amp_cache, fallbacks
from .amp_cache import lookup
from .ops.functions import attn_smallk_func

HIP_SMALLK = True       # A/B switch: csrc/attn_smallk.hip for the short-key shapes (text gate, decoder self-attention)


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

    def forward_batch_first(self, query, key, value, key_padding_mask=None):
        """query [B, Lq, C] (batch-first), key / value [Lk, B, C] -> [B, Lq, C], or None when the short-key HIP kernel does not serve the
        call (the caller then takes forward() on the token-major layout).  Same arithmetic per token as forward()."""
        B, Lq, C = query.shape
        H, hd = self.num_heads, C // self.num_heads
        if not (HIP_SMALLK and query.is_cuda and hd == 32 and key.shape[0] <= 32 and H <= 8 and not (self.training and self.dropout > 0)):
            return None
        w, b = lookup(self.in_proj_weight), lookup(self.in_proj_bias)
        wq, wk, wv = w.chunk(3)
        bq, bk, bv = b.chunk(3)
        q, k, v = amp_cache.linear(query, wq, bq), amp_cache.linear(key, wk, bk), amp_cache.linear(value, wv, bv)
        o = attn_smallk_func.attention_batch_first(q, k, v, key_padding_mask, hd ** -0.5, H)
        return None if o is None else self.out_proj(o)

    def forward(self, query, key, value, key_padding_mask=None):
        """query [Lq,B,C], key/value [Lk,B,C], key_padding_mask [B,Lk] (True = ignore) -> [Lq,B,C]."""
        Lq, B, C = query.shape
        Lk = key.shape[0]
        H, hd = self.num_heads, C // self.num_heads
        w, b = lookup(self.in_proj_weight), lookup(self.in_proj_bias)
        if HIP_SMALLK and query.is_cuda and hd == 32 and Lk <= 32 and H <= 8:
            # short key sequence: the attention core is one HIP kernel each way on the projections' own [L, B, C] layout
            # (no head permutes, no additive-mask tensor, no flash-attention launch for <= 32 keys)
            if query is key:
                wqk, wv = w.split([2 * C, C])
                bqk, bv = b.split([2 * C, C])
                qk = amp_cache.linear(query, wqk, bqk)
                q, k = qk[..., :C], qk[..., C:]
            else:
                wq, wk, wv = w.chunk(3)
                bq, bk, bv = b.chunk(3)
                q, k = amp_cache.linear(query, wq, bq), amp_cache.linear(key, wk, bk)
            v = amp_cache.linear(value, wv, bv)
            o = attn_smallk_func.attention(q, k, v, key_padding_mask, hd ** -0.5, H, self.dropout if self.training else 0.0)
            if o is not None:
                return self.out_proj(o)
        fallbacks.note("MultiheadAttention", f"head_dim {hd}, {H} heads, {Lk} keys not served by csrc/attn_smallk.hip"
                       if not (hd == 32 and Lk <= 32 and H <= 8) else "attn_smallk declined the call", query)
        # split the packed projection ONCE (backward: one cat per parameter, not three zero-fill + copy + add chains)
        if query is key:        # decoder self-attention: q and k from the same input -> one GEMM
            wqk, wv = w.split([2 * C, C])
            bqk, bv = b.split([2 * C, C])
            qk = F.linear(query, wqk, bqk).view(Lq, B, 2, H, hd)
            q, k = qk[:, :, 0].permute(1, 2, 0, 3), qk[:, :, 1].permute(1, 2, 0, 3)
        else:
            wq, wk, wv = w.chunk(3)
            bq, bk, bv = b.chunk(3)
            q = F.linear(query, wq, bq).view(Lq, B, H, hd).permute(1, 2, 0, 3)
            k = F.linear(key, wk, bk).view(Lk, B, H, hd).permute(1, 2, 0, 3)
        v = F.linear(value, wv, bv).view(Lk, B, H, hd).permute(1, 2, 0, 3)
        mask = None
        if key_padding_mask is not None:
            cache = key_padding_mask.__dict__.setdefault("_ocpg_additive", {}) if hasattr(key_padding_mask, "__dict__") else {}
            hit = cache.get(q.dtype)
            if hit is None or hit[0] != key_padding_mask._version:
                # the same padding mask serves every level / layer of a forward: build its additive form once per content version
                hit = cache[q.dtype] = (key_padding_mask._version, torch.zeros((B, 1, 1, Lk), dtype=q.dtype, device=q.device).masked_fill(
                    key_padding_mask[:, None, None, :], float("-inf")))
            mask = hit[1]
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=self.dropout if self.training else 0.0)
        return self.out_proj(o.permute(2, 0, 1, 3).reshape(Lq, B, C))
