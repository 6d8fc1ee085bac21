"""Video-Swin (T / S / B) spatio-temporal backbone with per-frame multi-scale outputs.

Behaviour follows the reference's models/video_swin_transformer.py: WindowAttention3D :87-169, SwinTransformerBlock3D
:172-274, PatchMerging :277-312, compute_mask :316-329, BasicLayer :332-413, PatchEmbed3D :416-456,
VideoSwinTransformerBackbone :638-701, configs :739-785, Joiner :787-806.  Parameter / buffer names are identical
(`patch_embed.{proj,norm}`, `layers.N.blocks.M.{norm1,attn.{qkv,proj,relative_position_bias_table,
relative_position_index},norm2,mlp.{fc1,fc2}}`, `downsamples.N.{norm,reduction}`), so reference checkpoints load.

MI355X-first structure.  The reference makes 8 full-tensor copies around every attention (pad, roll, partition,
permute, reverse, roll back, crop) and materialises the [windows, heads, N, N] score tensor in HBM (232 MB per block at
stage 1 of Swin-T, 1.37 GB for Swin-B).  Here pad + cyclic shift + window partition are folded into ONE precomputed
token permutation (a single gather in, a single gather out; padded slots read a zero row), and the score matrix never
leaves the attention kernel: relative-position bias and the shift mask are merged into one additive [nW, heads, N, N]
table per (stage geometry) that the fused scaled-dot-product kernel consumes.
Quirks kept: the window is clamped to the clip size per axis and the shift zeroed on clamped axes (:71-84);
`relative_position_index[:N, :N]` is sliced for the clamped window (:151); the shift mask uses -100, not -inf (:328);
the 4th stage is kept when output_levels == 4; drop-path rate 0.2 linearly scaled over the blocks.
"""
from functools import lru_cache

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import amp_cache, fallbacks

from ..util.misc import NestedTensor, mask_key, resize_mask
from .ops.functions.win_attn_func import window_attention
from .ops.functions.layernorm_func import LayerNorm as _FusedLayerNorm, PermuteGather, RelPosBias, StaticGather, permute_gather_ok
from .position_encoding import build_position_encoding


import os
_GENERIC_ATTENTION = os.environ.get("OCPG_GENERIC_WINDOW_ATTENTION") == "1"      # A/B switch: torch SDPA instead of csrc/win_attn.hip


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask.div_(keep)          # timm scales the [B, 1, ...] mask, then one multiply over the map

    def add(self, res, x):
        """res + self(x) as ONE pass over the map when the path drop is live (addcmul; the separate multiply was 72 launches / 0.85 ms
        per config-#5 step; the product is formed in the sum's precision instead of being rounded to x's dtype first)."""
        if self.drop_prob == 0.0 or not self.training:
            return res + x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep).div_(keep)
        return torch.addcmul(res, x, mask)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = amp_cache.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = amp_cache.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


def get_window_size(x_size, window_size, shift_size=None):
    """Clamp the window to the clip extent per axis; a clamped axis is not shifted."""
    ws = [min(x, w) for x, w in zip(x_size, window_size)]
    if shift_size is None:
        return tuple(ws)
    ss = [0 if x <= w else s for x, w, s in zip(x_size, window_size, shift_size)]
    return tuple(ws), tuple(ss)


@lru_cache(maxsize=None)
def _window_plan(D, H, W, ws, ss):
    """Token permutation of one (geometry, window, shift): for every window slot the index of its source token in the
    UNPADDED [D*H*W] sequence (or D*H*W for a padded slot, which reads an appended zero row), the inverse gather for the
    way back, and the shift-mask region id of every slot.  numpy, computed once per geometry."""
    Dp, Hp, Wp = (int(np.ceil(x / w)) * w for x, w in zip((D, H, W), ws))
    d, h, w = np.meshgrid(np.arange(Dp), np.arange(Hp), np.arange(Wp), indexing="ij")
    # shifted_x[i] = x_padded[(i + shift) mod size]  (torch.roll by -shift)
    sd, sh, sw = (d + ss[0]) % Dp, (h + ss[1]) % Hp, (w + ss[2]) % Wp
    src = np.where((sd < D) & (sh < H) & (sw < W), (sd * H + sh) * W + sw, D * H * W)
    # region ids of compute_mask (:316-329): 3 slabs per axis on the SHIFTED grid

    def slab(n, wsz, ssz):
        r = np.zeros(n, dtype=np.int64)
        if ssz > 0:
            r[n - wsz:n - ssz] = 1
            r[n - ssz:] = 2
        return r
    # NB: with shift 0 the reference's slices (slice(-w), slice(-w, -0), slice(-0, None)) leave every position in
    # the LAST slab of that axis; only differences matter, so 0 is equivalent.
    region = slab(Dp, ws[0], ss[0])[d] * 9 + slab(Hp, ws[1], ss[1])[h] * 3 + slab(Wp, ws[2], ss[2])[w]

    def partition(a):
        a = a.reshape(Dp // ws[0], ws[0], Hp // ws[1], ws[1], Wp // ws[2], ws[2])
        return a.transpose(0, 2, 4, 1, 3, 5).reshape(-1, ws[0] * ws[1] * ws[2])
    src_w = partition(src)                                    # [nW, N]
    region_w = partition(region)
    # inverse: for each original token, which window slot holds it
    inv = np.full(D * H * W + 1, -1, dtype=np.int64)
    flat = src_w.reshape(-1)
    inv[flat] = np.arange(flat.size)
    return src_w, inv[:-1], region_w, (Dp, Hp, Wp)


def gather_in_inv(gather_in, S, owner, key):
    """Backward map of the reverse gather: for every window SLOT the token it holds, or -1 for a padded slot = gather_in with the
    out-of-range marker replaced (cached on the block)."""
    cache = owner.__dict__.setdefault("_gather_inv", {})
    k = (key, gather_in.device)
    if k not in cache:
        cache[k] = torch.where(gather_in < S, gather_in, torch.full_like(gather_in, -1))
    return cache[k]


def _lp_norm(norm_layer, dim):
    """The norms that feed a Linear (norm1, norm2, PatchMerging.norm): the fused low-precision LayerNorm when the stock one was asked for."""
    return _FusedLayerNorm(dim) if norm_layer is nn.LayerNorm and _FUSED_LN else norm_layer(dim)


_FUSED_LN = os.environ.get("OCPG_FUSED_SWIN_LN", "1") != "0"
_RELPOS_KERNEL = os.environ.get("OCPG_RELPOS_KERNEL", "1") != "0"     # A/B switch: relative-position bias (both layouts) from the table in one launch


class WindowAttention3D(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wd, wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += wd - 1
        rel[:, :, 1] += wh - 1
        rel[:, :, 2] += ww - 1
        rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
        rel[:, :, 1] *= (2 * ww - 1)
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = amp_cache.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = amp_cache.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def relative_position_bias(self, n):
        """[heads, N, N]; the index table is sliced [:N, :N] for a clamped window exactly as the reference does."""
        table = self.relative_position_bias_table
        self._bias_t = None
        if _RELPOS_KERNEL and table.is_cuda and table.dtype == torch.float32 and table.is_contiguous():
            # both layouts the attention kernels read, from the table, in one launch (csrc/layernorm.hip relpos_bias)
            idx2 = self.relative_position_index[:n, :n]
            plan = self.__dict__.get("_relpos_plan")
            if plan is None or plan[0] != (n, table.device):
                plan = self.__dict__["_relpos_plan"] = ((n, table.device), StaticGather.plan(idx2.reshape(-1), table.shape[0]))
            bias, self._bias_t = RelPosBias.apply(table, idx2, plan[1])
            return bias
        idx = self.relative_position_index[:n, :n].reshape(-1)
        if table.is_cuda and table.dtype == torch.float32 and table.requires_grad and torch.is_grad_enabled():
            plan = self.__dict__.get("_gather_plan")                     # the index is a buffer: sorted by destination once per N
            if plan is None or plan[0] != (n, table.device):
                plan = self.__dict__["_gather_plan"] = ((n, table.device), StaticGather.plan(idx, table.shape[0]))
            return StaticGather.apply(table, idx, plan[1]).view(n, n, -1).permute(2, 0, 1)
        return table[idx].view(n, n, -1).permute(2, 0, 1)

    def forward(self, x, mask=None, region=None):
        """x [num_windows*B, N, C]; mask [num_windows, N, N] additive (0 / -100) or None; region [num_windows, N] int32 =
        the same shift mask as region ids (what the fused HIP kernel consumes instead of the N x N tensor)."""
        bw, n, c = x.shape
        h = self.num_heads
        qkv = self.qkv(x)
        bias = self.relative_position_bias(n)                                            # [h, N, N]
        if (x.is_cuda and c // h == 32 and (self.attn_drop.p == 0.0 or not self.training) and (mask is None or region is not None)
                and not _GENERIC_ATTENTION):
            nw = region.shape[0] if region is not None else 1
            out = window_attention(qkv.view(bw, n, 3, h, c // h), bias, region, self.scale, nw, self.__dict__.pop("_bias_t", None))      # csrc/win_attn.hip
            return self.proj_drop(self.proj(out))
        # generic path (CPU unit tests of the host logic, head_dim != 32, attention dropout): torch's fused SDPA
        fallbacks.note("WindowAttention3D", "generic attention forced (OCPG_GENERIC_WINDOW_ATTENTION)" if _GENERIC_ATTENTION else
                       f"head_dim {c // h} / attention dropout {self.attn_drop.p} / dense shift mask not served by csrc/win_attn*.hip", x)
        qkv = qkv.view(bw, n, 3, h, c // h).permute(2, 0, 3, 1, 4)
        bias = bias.unsqueeze(0)
        if mask is None and region is not None:          # the additive N x N form of the same shift mask
            r = region.long()
            mask = torch.zeros((r.shape[0], n, n), dtype=torch.float32, device=x.device).masked_fill(r[:, None, :] != r[:, :, None], -100.0)
        if mask is not None:
            nw = mask.shape[0]
            bias = (bias + mask.unsqueeze(1)).unsqueeze(0).expand(bw // nw, nw, h, n, n).reshape(bw, h, n, n)
        out = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=bias.to(qkv.dtype),
                                             dropout_p=self.attn_drop.p if self.training else 0.0, scale=self.scale)
        return self.proj_drop(self.proj(out.transpose(1, 2).reshape(bw, n, c)))


class SwinTransformerBlock3D(nn.Module):
    def __init__(self, dim, num_heads, window_size=(2, 7, 7), shift_size=(0, 0, 0), mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 use_checkpoint=False):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.window_size, self.shift_size = tuple(window_size), tuple(shift_size)
        self.mlp_ratio, self.use_checkpoint = mlp_ratio, use_checkpoint
        assert all(0 <= s < w for s, w in zip(self.shift_size, self.window_size)), "shift_size must in 0-window_size"
        self.norm1 = _lp_norm(norm_layer, dim)
        self.attn = WindowAttention3D(dim, window_size=self.window_size, num_heads=num_heads, qkv_bias=qkv_bias,
                                      qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = _lp_norm(norm_layer, dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if not use_checkpoint:      # every Linear of a block runs once per forward: its gradient sums ride in the fused gradient cast
            amp_cache.mark_single_use(self.attn.qkv, self.attn.proj, self.mlp.fc1, self.mlp.fc2)

    def _plan(self, D, H, W, device):
        ws, ss = get_window_size((D, H, W), self.window_size, self.shift_size)
        cache = self.__dict__.setdefault("_plans", {})
        key = (D, H, W, ws, ss, str(device))
        if key not in cache:
            src_w, inv, region_w, _ = _window_plan(D, H, W, ws, ss)
            mask = region = None
            if any(s > 0 for s in ss):
                r = torch.from_numpy(region_w)
                region = r.to(torch.int32).contiguous().to(device)
                if device.type != "cuda":        # the N x N additive mask only exists for the generic (SDPA) path
                    diff = r[:, None, :] - r[:, :, None]
                    mask = torch.zeros(diff.shape, dtype=torch.float32).masked_fill(diff != 0, -100.0).to(device)
            cache[key] = (torch.from_numpy(src_w.reshape(-1)).to(device), torch.from_numpy(inv).to(device), mask, region, src_w.shape)
        return cache[key]

    def forward_part1(self, x, mask_matrix=None):
        B, D, H, W, C = x.shape
        gather_in, gather_out, mask, region, (nw, n) = self._plan(D, H, W, x.device)
        x = self.norm1(x).view(B, D * H * W, C)
        if permute_gather_ok(x):
            # pad + roll + partition as ONE gather kernel whose backward is the reverse gather (every token sits in exactly one slot:
            # no index_add atomics, no appended zero row); gather_in marks padded slots with D*H*W (out of range -> zeros)
            windows = PermuteGather.apply(x, gather_in, gather_out).view(B * nw, n, C)
            out = self.attn(windows, mask=mask, region=region).view(B, nw * n, C)
            if permute_gather_ok(out):
                return PermuteGather.apply(out, gather_out, gather_in_inv(gather_in, D * H * W, self, (D, H, W))).view(B, D, H, W, C)
            return out.index_select(1, gather_out).view(B, D, H, W, C)
        x = torch.cat([x, x.new_zeros(B, 1, C)], dim=1)                                   # the zero row read by padded slots
        windows = x.index_select(1, gather_in).view(B * nw, n, C)                         # pad + roll + partition: one gather
        out = self.attn(windows, mask=mask, region=region).view(B, nw * n, C)
        return out.index_select(1, gather_out).view(B, D, H, W, C)                        # reverse + un-roll + crop: one gather

    def forward_part2(self, x):
        return self.drop_path(self.mlp(self.norm2(x)))

    def forward(self, x, mask_matrix=None):
        if self.use_checkpoint:
            from torch.utils.checkpoint import checkpoint
            x = x + self.drop_path(checkpoint(self.forward_part1, x, mask_matrix, use_reentrant=False))
            return x + checkpoint(self.forward_part2, x, use_reentrant=False)
        add = self.drop_path.add if isinstance(self.drop_path, DropPath) else (lambda r, y: r + y)
        x = add(x, self.forward_part1(x, mask_matrix))
        return add(x, self.mlp(self.norm2(x)))


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = amp_cache.Linear(4 * dim, 2 * dim, bias=False)
        amp_cache.mark_single_use(self.reduction)
        self.norm = _lp_norm(norm_layer, 4 * dim)

    def forward(self, x):
        """[B, D, H, W, C] -> [B, D, ceil(H/2), ceil(W/2), 2C]; channel order (0,0), (1,0), (0,1), (1,1)."""
        B, D, H, W, C = x.shape
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
        return self.reduction(self.norm(x))


def compute_mask(D, H, W, window_size, shift_size, device):
    """Shift mask [nW, N, N] (0 / -100) of a PADDED D x H x W grid -- same values as the reference's compute_mask."""
    _, _, region_w, _ = _window_plan(D, H, W, tuple(window_size), tuple(shift_size))
    r = torch.from_numpy(region_w)
    diff = r[:, None, :] - r[:, :, None]
    return torch.zeros(diff.shape, dtype=torch.float32).masked_fill(diff != 0, -100.0).to(device)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=(1, 7, 7), mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        self.depth, self.use_checkpoint = depth, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim=dim, num_heads=num_heads, window_size=window_size,
                                   shift_size=(0, 0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio=mlp_ratio,
                                   qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                   drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                   norm_layer=norm_layer, use_checkpoint=use_checkpoint) for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        """[B, C, D, H, W] -> [B, C', D, H', W']."""
        x = x.permute(0, 2, 3, 4, 1)
        if not x.is_contiguous():      # (the previous stage / the patch embedding hand over channels-last memory: then this is a view --
            x = x.contiguous()         #  a [B, C, D, H, W]-contiguous input made every op of the first block a strided one: 512 us per add at config #5)
        for blk in self.blocks:
            x = blk(x)
        if self.downsample is not None:
            x = self.downsample(x)
        return x.permute(0, 4, 1, 2, 3)


class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size=(2, 4, 4), in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size, self.in_chans, self.embed_dim = tuple(patch_size), in_chans, embed_dim
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = _lp_norm(norm_layer, embed_dim) if norm_layer is not None else None
        if isinstance(self.norm, _FusedLayerNorm):
            self.norm.fp32_out = True            # this norm's output IS the residual stream: fp32, as autocast's layer_norm returns it

    def forward(self, x):
        _, _, D, H, W = x.shape
        pd, ph, pw = self.patch_size
        if W % pw or H % ph or D % pd:
            x = F.pad(x, (0, (pw - W % pw) % pw, 0, (ph - H % ph) % ph, 0, (pd - D % pd) % pd))
        if pd == 1:
            # temporal patch size 1 (every Video-Swin config of the reference): the Conv3d is a per-frame 2-D conv.
            # Same weights ([C, 3, 1, ph, pw], checkpoint-compatible), but MIOpen's 2-D path instead of its 3-D solvers.
            b, c, d, h, w = x.shape
            xin = x.transpose(1, 2).reshape(b * d, c, h, w)
            if xin.is_cuda:                      # channels-last in -> channels-last out: the token view below is then contiguous (no copy into the norm)
                xin = xin.contiguous(memory_format=torch.channels_last)
            y = F.conv2d(xin, self.proj.weight.squeeze(2), self.proj.bias, stride=(ph, pw))
            x = y.view(b, d, self.embed_dim, y.shape[-2], y.shape[-1]).transpose(1, 2)
        else:
            x = self.proj(x)
        if self.norm is not None:
            x = self.norm(x.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)
        return x


class SwinTransformer3D(nn.Module):
    def __init__(self, pretrained=None, pretrained2d=True, patch_size=(4, 4, 4), in_chans=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=(2, 7, 7), mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.2, norm_layer=nn.LayerNorm, patch_norm=False, frozen_stages=-1,
                 use_checkpoint=False):
        super().__init__()
        self.num_layers, self.embed_dim, self.patch_norm = len(depths), embed_dim, patch_norm
        self.frozen_stages, self.window_size, self.patch_size = frozen_stages, window_size, patch_size
        self.patch_embed = PatchEmbed3D(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                        norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                                          mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                                          attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                          norm_layer=norm_layer, downsample=PatchMerging if i < self.num_layers - 1 else None,
                                          use_checkpoint=use_checkpoint))
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.norm = norm_layer(self.num_features)


class VideoSwinTransformerBackbone(nn.Module):
    """Per-frame outputs of every stage BEFORE its patch merging ((b t), C_i, H/2^(i+2), W/2^(i+2))."""

    def __init__(self, backbone_pretrained, backbone_pretrained_path, train_backbone, out_indices, **kwargs):
        super().__init__()
        swin = SwinTransformer3D(**kwargs)
        if backbone_pretrained and isinstance(backbone_pretrained_path, str):
            sd = torch.load(backbone_pretrained_path, map_location="cpu")["state_dict"]
            sd = {k[9:]: v for k, v in sd.items() if "backbone." in k}
            sd["patch_embed.proj.weight"] = sd["patch_embed.proj.weight"].sum(dim=2, keepdims=True)
            swin.load_state_dict(sd)
        self.patch_embed, self.pos_drop = swin.patch_embed, swin.pos_drop
        self.layers = swin.layers if len(out_indices) != 3 else swin.layers[:-1]
        self.downsamples = nn.ModuleList()
        for layer in self.layers:
            self.downsamples.append(layer.downsample)
            layer.downsample = None
        self.downsamples[-1] = None
        self.layer_output_channels = [swin.embed_dim * 2 ** i for i in range(len(self.layers))]
        self.train_backbone = train_backbone
        if not train_backbone:
            for p in self.parameters():
                p.requires_grad_(False)

    def forward(self, samples, num_frames):
        n, c, h, w = samples.shape
        x = samples.view(n // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.pos_drop(self.patch_embed(x))
        out = {}
        for idx, (layer, down) in enumerate(zip(self.layers, self.downsamples)):
            x = layer(x)               # (no .contiguous(): the stages work on [B, D, H, W, C] memory, which is what arrives here)
            out[str(idx)] = x
            if down is not None:
                x = down(x.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)
        return {k: v.permute(0, 2, 1, 3, 4).flatten(0, 1) for k, v in out.items()}


configs = {
    "video_swin_t_p4w7": dict(patch_size=(1, 4, 4), embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=(8, 7, 7),
                              mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2,
                              patch_norm=True, use_checkpoint=False),
    "video_swin_s_p4w7": dict(patch_size=(1, 4, 4), embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=(8, 7, 7),
                              mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2,
                              patch_norm=True, use_checkpoint=False),
    "video_swin_b_p4w7": dict(patch_size=(1, 4, 4), embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=(8, 7, 7),
                              mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2,
                              patch_norm=True, use_checkpoint=False),
}


class BackboneBase(nn.Module):
    def __init__(self, backbone, strides=(4, 8, 16, 32), num_channels=(96, 192, 384, 768)):
        super().__init__()
        self.strides, self.num_channels = list(strides), list(num_channels)
        self.body = backbone

    def forward(self, tensor_list: NestedTensor, num_frames: int):
        m = tensor_list.mask
        assert m is not None
        out = {}
        for name, x in self.body(tensor_list.tensors, num_frames).items():
            out[name] = NestedTensor(x, resize_mask(m, x.shape[-2:]))        # memoised on the mask's valid extents, like the ResNet Joiner's
        return out


class Backbone(BackboneBase):
    def __init__(self, name, checkpoint=False, pretrained=None, output_levels=4, cfg_override=None):
        assert name in configs
        cfg = dict(configs[name])
        cfg.update(cfg_override or {})
        cfg["use_checkpoint"] = checkpoint
        out_indices = tuple(range(output_levels))
        super().__init__(VideoSwinTransformerBackbone(True, pretrained, True, out_indices, **cfg),
                         [2 ** (i + 2) for i in out_indices], [int(cfg["embed_dim"] * 2 ** i) for i in out_indices])


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides, self.num_channels = backbone.strides, backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        t = tensor_list.tensors.shape[1]
        tensor_list.tensors = tensor_list.tensors.flatten(0, 1)          # NB: folds the caller's NestedTensor in place
        key = mask_key(tensor_list.mask)
        tensor_list.mask = tensor_list.mask.flatten(0, 1)
        if key is not None:             # (the tag of the padding mask's valid extents survives the fold: level masks and position
            tensor_list.mask._ocpg_key = key      #  encodings are memoised on it, as in models/backbone.py's Joiner)
        xs = self[0](tensor_list, num_frames=t)
        out = [x for _, x in sorted(xs.items())]
        return out, [self[1](x).to(x.tensors.dtype) for x in out]


def build_video_swin_backbone(args):
    backbone = Backbone(args.backbone, args.use_checkpoint, args.backbone_pretrained, args.output_levels,
                        cfg_override=getattr(args, "video_swin_cfg", None))
    return Joiner(backbone, build_position_encoding(args))
