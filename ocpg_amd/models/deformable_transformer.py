"""Multi-scale deformable cross-modal transformer (4 encoder + 4 decoder layers by default).

Behaviour follows the reference's models/deformable_transformer.py: DeformableTransformer.forward :134-217,
encoder layer :220-259, get_reference_points :268-281, decoder layer :292-336 (self-attn -> cross-attn -> FFN with
norms named norm2 / norm1 / norm3), decoder :340-398 (top-30 sample export :368-375, iterative box refinement
:378-388).  Parameter names are identical so reference state_dicts load.

Host-side differences (MI355X-first): level geometry (shapes, level starts) is kept as Python ints next to the
device tensors (`_ocpg_host`), so no kernel launch depends on a device->host read; the reference point grid and
the per-level slices are built from those ints.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn

from ..util.misc import fully_valid, inverse_sigmoid, mask_key, memo
from . import amp_cache
from .attention import MultiheadAttention
from .ops.functions import fused_ln_func
from .ops.modules import MSDeformAttn

FUSED_GLUE = True       # A/B switch: fused dropout+add+LayerNorm and bias+ReLU+dropout kernels (csrc/fused_ln.hip)


def _drop_add_norm(x, res, drop, norm):
    """norm(res + drop(x)): one fused HIP pass each way on the GPU (LayerNorm over <= 2048 channels, fp32 residual stream)."""
    if FUSED_GLUE and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and fused_ln_func.supported(x, res, x.shape[-1]):
        return fused_ln_func.dropout_add_layer_norm(x, res, norm, drop.p if drop.training else 0.0)
    return norm(res + drop(x))


def _ffn_hidden(src, linear1, activation, drop):
    """drop(activation(linear1(src))): GEMM + one fused bias/ReLU/dropout pass on the GPU (ReLU FFNs)."""
    if FUSED_GLUE and activation is F.relu and src.is_cuda and linear1.bias is not None:
        w, b = amp_cache.lookup(linear1.weight), amp_cache.lookup(linear1.bias)
        x = src
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x, w, b = x.to(dt), w.to(dt), b.to(dt)
        if x.dtype == w.dtype and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and w.shape[0] % 4 == 0:
            x2 = x.reshape(-1, x.shape[-1])
            splits = amp_cache._split_rows(x2.shape[0]) if amp_cache.SPLIT_K else 1
            h = fused_ln_func.LinearBiasReluDropout.apply(x2, w, b, drop.p if drop.training else 0.0, None, splits)
            return h.view(*x.shape[:-1], w.shape[0])
    return drop(activation(linear1(src)))


def _clones(module, n):
    """n deep copies (deformable_transformer.py:401-402).  Parameter.__deepcopy__ drops instance attributes, so the single-use marks
    (amp_cache.mark_single_use: deferred weight-gradient sums) are carried over to the copies' parameters by hand."""
    out = []
    for _ in range(n):
        c = copy.deepcopy(module)
        for p, q in zip(module.parameters(), c.parameters()):
            if getattr(p, "_ocpg_single_use", False):
                q._ocpg_single_use = True
        out.append(c)
    return nn.ModuleList(out)


def _activation(name):
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    if name == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {name}.")


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = amp_cache.Linear(d_model, d_ffn)
        amp_cache.mark_single_use(self.linear1)
        self.activation = _activation(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = amp_cache.Linear(d_ffn, d_model)
        amp_cache.mark_single_use(self.linear2)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        with torch.autocast(device_type=src.device.type, enabled=False):
            q = src.float() if pos is None else src.float() + pos.float()
            attn = self.self_attn(q, reference_points, src.float(), spatial_shapes, level_start_index, padding_mask)[0]
        src = _drop_add_norm(attn, src, self.dropout1, self.norm1)
        ffn = self.linear2(_ffn_hidden(src, self.linear1, self.activation, self.dropout2))
        return _drop_add_norm(ffn, src, self.dropout3, self.norm2)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel-centre grid of every level, normalised by the valid extent, re-scaled per target level: [N, S, L, 2]."""
        host = getattr(spatial_shapes, "_ocpg_host", None)
        shapes = host.tolist() if host is not None else spatial_shapes.tolist()
        per_level = []
        for lvl, (h, w) in enumerate(shapes):
            ys = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32, device=device)
            xs = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32, device=device)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            gy = gy.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            gx = gx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            per_level.append(torch.stack((gx, gy), -1))
        ref = torch.cat(per_level, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None):
        ref = memo("enc_ref_points", getattr(valid_ratios, "_ocpg_key", None), src.device,
                   lambda: self.get_reference_points(spatial_shapes, valid_ratios, src.device))
        out = src
        for layer in self.layers:
            out = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask)
        return out


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = amp_cache.Linear(d_model, d_ffn)
        amp_cache.mark_single_use(self.linear1)
        self.activation = _activation(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = amp_cache.Linear(d_ffn, d_model)
        amp_cache.mark_single_use(self.linear2)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index, src_padding_mask=None):
        qk = tgt if query_pos is None else tgt + query_pos
        sa = self.self_attn(qk.transpose(0, 1), qk.transpose(0, 1), tgt.transpose(0, 1)).transpose(0, 1)
        tgt = _drop_add_norm(sa, tgt, self.dropout2, self.norm2)
        with torch.autocast(device_type=tgt.device.type, enabled=False):
            q = tgt.float() if query_pos is None else tgt.float() + query_pos.float()
            ca, loc, weights = self.cross_attn(q, reference_points, src.float(), src_spatial_shapes, level_start_index, src_padding_mask)
        tgt = _drop_add_norm(ca, tgt, self.dropout1, self.norm1)
        ffn = self.linear2(_ffn_hidden(tgt, self.linear1, self.activation, self.dropout3))
        return _drop_add_norm(ffn, tgt, self.dropout4, self.norm3), loc, weights


class DeformableTransformerDecoder(nn.Module):
    TOPK_SAMPLES = 30   # deformable_transformer.py:372-373

    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        # the top-k sampling locations of every layer (deformable_transformer.py:368-375,393): returned by the reference's
        # transformer and never read by its model (models/ocpg.py:300 is the only consumer and drops them) -- OCPG switches the
        # computation (div + topk + gather per layer) off; the decoder on its own keeps the reference's behaviour
        self.compute_samples = True
        self.bbox_embed = None      # set by OCPG for iterative box refinement
        self.class_embed = None

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None):
        out = tgt
        inter, inter_refs, inter_samples = [], [], []
        self.box_deltas = [] if self.bbox_embed is not None else None      # handed to the caller's box head (same module, same input)
        samples_keep = None
        for lid, layer in enumerate(self.layers):
            if reference_points.shape[-1] == 4:
                ref_in = reference_points[:, :, None] * torch.cat([src_valid_ratios, src_valid_ratios], -1)[:, None]
            else:
                assert reference_points.shape[-1] == 2
                ref_in = reference_points[:, :, None] * src_valid_ratios[:, None]
            out, loc, weights = layer(out, query_pos, ref_in, src, src_spatial_shapes, src_level_start_index, src_padding_mask)

            if self.compute_samples:
                n, lq = loc.shape[:2]
                loc = loc / src_valid_ratios[:, None, None, :, None, :]
                top_idx = weights.reshape(n, lq, -1).topk(self.TOPK_SAMPLES, dim=2)[1]
                samples_keep = torch.gather(loc.reshape(n, lq, -1, 2), 2, top_idx.unsqueeze(-1).expand(-1, -1, -1, 2))

            if self.bbox_embed is not None:
                delta = self.bbox_embed[lid](out)
                self.box_deltas.append(delta)
                if reference_points.shape[-1] == 4:
                    new_ref = (delta + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    new_ref = (delta + F.pad(inverse_sigmoid(reference_points), (0, 2))).sigmoid()
                reference_points = new_ref.detach()
            if self.return_intermediate:
                inter.append(out)
                inter_refs.append(reference_points)
                if self.compute_samples:
                    inter_samples.append(samples_keep)
        if self.return_intermediate:
            return torch.stack(inter), torch.stack(inter_refs), (torch.stack(inter_samples) if self.compute_samples else None)
        return out, reference_points, samples_keep


class _LevelPos(torch.autograd.Function):
    """cat_l(pos_l) + level_embed[l] broadcast over the tokens of level l (deformable_transformer.py:158-159); pos is a constant."""

    @staticmethod
    def forward(ctx, pos_cat, level_embed, sizes):
        ctx.sizes = sizes
        out = pos_cat.clone()
        start = 0
        for l, n in enumerate(sizes):
            out[:, start:start + n] += level_embed[l]
            start += n
        return out

    @staticmethod
    def backward(ctx, g):
        gs = g.sum(0)                                    # [S, C]: one pass over the frames
        parts, start = [], 0
        for n in ctx.sizes:
            parts.append(gs[start:start + n].sum(0))
            start += n
        return None, torch.stack(parts).to(g.dtype), None


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=300):
        super().__init__()
        assert not two_stage, "args.two_stage must be false!"   # ocpg.py:65
        self.d_model, self.nhead, self.dropout = d_model, nhead, dropout
        self.two_stage = False
        self.num_feature_level = num_feature_levels
        self.encoder = DeformableTransformerEncoder(
            DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, enc_n_points),
            num_encoder_layers)
        self.decoder = DeformableTransformerDecoder(
            DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, dec_n_points),
            num_decoder_layers, return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.empty(num_feature_levels, d_model))
        self.reference_points = amp_cache.Linear(d_model, 2)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.xavier_uniform_(self.reference_points.weight, gain=1.0)
        nn.init.zeros_(self.reference_points.bias)
        nn.init.normal_(self.level_embed)

    def _level_geometry(self, shapes_host, dev):
        """Device copies of (spatial_shapes, level_start_index) with their host twins attached, cached per geometry:
        no host->device copy (and so nothing un-capturable by a HIP graph) happens after the first step."""
        cache = self.__dict__.setdefault("_geom_cache", {})
        key = (shapes_host, str(dev))
        if key not in cache:
            host = torch.tensor(shapes_host, dtype=torch.long)
            starts_host = torch.cat((host.new_zeros(1), host.prod(1).cumsum(0)[:-1]))
            spatial_shapes, level_start_index = host.to(dev), starts_host.to(dev)
            spatial_shapes._ocpg_host = host
            level_start_index._ocpg_host = starts_host
            cache[key] = (spatial_shapes, level_start_index)
        return cache[key]

    @staticmethod
    def get_valid_ratio(mask):
        _, h, w = mask.shape
        vh = (~mask[:, :, 0]).sum(1).float() / h
        vw = (~mask[:, 0, :]).sum(1).float() / w
        return torch.stack([vw, vh], -1)

    def forward(self, srcs, tgt, masks, pos_embeds, query_embed=None):
        """srcs/pos_embeds: L x [(b t), C, h, w]; masks: L x [(b t), h, w]; tgt [b, t, q, C]; query_embed [q, C]."""
        assert query_embed is not None
        dev = srcs[0].device
        shapes_host = [tuple(s.shape[-2:]) for s in srcs]
        src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        # no padding anywhere (known on the host from the collate step): masked_fill(value, all-False) is the identity -- skip it
        # and its backward in every MSDeformAttn call (2 x 52 MB passes each at config #2)
        unpadded = all(fully_valid(mask_key(m)) for m in masks)
        mask = None if unpadded else torch.cat([m.flatten(1) for m in masks], 1)
        if all(not p.requires_grad for p in pos_embeds):
            # position encodings are constants: one add forward, and backward ONE sum over the frames + L small segment sums for
            # level_embed (the per-level broadcast adds cost 4 x (80 us reduction + zeros + copy + add) at config #2)
            pos = _LevelPos.apply(torch.cat([p.flatten(2).transpose(1, 2) for p in pos_embeds], 1), self.level_embed,
                                  tuple(h * w for h, w in shapes_host))
        else:
            pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[l].view(1, 1, -1) for l, p in enumerate(pos_embeds)], 1)
        spatial_shapes, level_start_index = self._level_geometry(tuple(shapes_host), dev)
        keys = tuple(mask_key(m) for m in masks)
        keys = None if any(k is None for k in keys) else keys
        valid_ratios = memo("valid_ratios", keys, dev, lambda: torch.stack([self.get_valid_ratio(m) for m in masks], 1))
        if keys is not None:
            valid_ratios._ocpg_key = keys

        memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, pos, mask)

        b, t, q, c = tgt.shape
        tgt = tgt.reshape(b * t, q, c)
        query_pos = query_embed[None].expand(b * t, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        hs, inter_refs, inter_samples = self.decoder(tgt, reference_points, memory, spatial_shapes, level_start_index,
                                                     valid_ratios, query_pos, mask)
        # with iterative box refinement the decoder has already evaluated bbox_embed[l](hs[l]) (only a detached copy is used for the
        # next layer's reference points): the model's box head re-uses these instead of running the same three GEMMs again.
        # Consumed here so that no autograd graph stays referenced from the module between steps.
        box_deltas, self.decoder.box_deltas = (self.decoder.box_deltas if self.decoder.return_intermediate else None), None
        feats, start = [], 0
        for (h, w) in shapes_host[: self.num_feature_level - 1]:
            feats.append(memory[:, start:start + h * w].reshape(b * t, h, w, c).permute(0, 3, 1, 2).contiguous())
            start += h * w
        return hs, feats, reference_points, inter_refs, box_deltas, None, inter_samples


def build_deforamble_transformer(args):
    return DeformableTransformer(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        activation="relu", return_intermediate_dec=True, num_feature_levels=args.num_feature_levels,
        dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points, two_stage=args.two_stage,
        two_stage_num_proposals=args.num_queries)
