"""OCPG model: backbone -> spectrum-guided text fusion neck -> deformable transformer -> box/class heads ->
dynamic-conv mask head (CPK) -> in-forward matching -> MSO refinement.

API and behaviour follow the reference's models/ocpg.py (`OCPG.__init__` :37-137, `forward` :197-447,
`dynamic_mask_with_coords` :475-529, `mask_heads_forward` :531-549, `parse_dynamic_params` :552-569,
`compute_locations` :596-609, `MLP` :613-625, `build` :635-718); parameter / buffer names are identical so that
reference checkpoints load and main.py's name-based LR groups select the same tensors.

MI355X-first differences in HOW (results agree to fp32 rounding):
  * the dynamic mask head never materialises the [1, b*t*q*258, h, w] repeat+cat tensor (99 MB per call in the
    reference at config #2): layer 1 is one batched contraction of the 256 feature channels for all queries of a
    frame plus a rank-2 closed form for the two relative-coordinate channels; layer 2 is a batched 16x16 product;
  * the per-decoder-layer loops (heads, controller, dynamic conv, matcher, MSO) are batched over layers;
  * MSO's 3x3 convs over the (layer-independent) backbone features are evaluated once and shared by the layers;
  * no host synchronisation inside forward (target sizes, matcher and index selection stay on the device).
"""
import copy
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..util.misc import NestedTensor, inverse_sigmoid, nested_tensor_from_videos_list, resize_mask
from . import amp_cache
from .backbone import build_backbone
from .criterion import SetCriterion
from .decoder import MSO
from .deformable_transformer import build_deforamble_transformer
from .matcher import build_matcher
from .modules import LFMResizeAdaptive
from .ops.functions.dynmask_func import dynamic_mask
from .ops.functions.groupnorm_func import GroupNorm
from .position_encoding import PositionEmbeddingSine1D
from .postprocessors import build_postprocessors
from .resample import bicubic_resize, bilinear_resize, nearest_upsample

MATMUL_BILINEAR = True       # A/B switch: bilinear resampling as two matrix products (no atomics backward)
CL_FUSE = os.environ.get("OCPG_CL_FUSE", "1") != "0"         # A/B switch: the fused visual map leaves the text gate in channels-last memory
GATE_BF = os.environ.get("OCPG_GATE_BF", "0") != "0"         # opt-in: the text gate on the batch-first token view of the channels-last map (no token-major round trip; measured 0.4 ms/step SLOWER, r4)
LS_FEAT_N16 = os.environ.get("OCPG_LS_FEAT_N16", "1") != "0"     # A/B switch: ls_feat_viz (3x3, 256 -> 8) by csrc/mso.hip's <= 16-output kernels
from .segmentation import VisionLanguageFusionModule
from .text_encoder.text_encoder import FeatureResizer, PrecomputedText, TextEncoder


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class MLP(nn.Module):
    """ReLU MLP (FFN) with `num_layers` Linear layers."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(amp_cache.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            if i < self.num_layers - 1:
                x = amp_cache.linear_relu(x, amp_cache.lookup(layer.weight), None if layer.bias is None else amp_cache.lookup(layer.bias))
            else:
                x = layer(x)
        return x


def compute_locations(h, w, device, stride=1):
    """Pixel-centre coordinates (x, y) of an h x w grid at `stride`, row-major, [h*w, 2]."""
    xs = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device) + stride // 2
    ys = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device) + stride // 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1)


def parse_dynamic_params(params, channels, weight_nums, bias_nums):
    """Split [n, sum(w)+sum(b)] controller outputs into per-layer conv weights [(n*channels), cin, 1, 1] and biases."""
    assert params.dim() == 2 and len(weight_nums) == len(bias_nums)
    assert params.size(1) == sum(weight_nums) + sum(bias_nums)
    n, nl = params.size(0), len(weight_nums)
    parts = list(torch.split_with_sizes(params, weight_nums + bias_nums, dim=1))
    weights = [parts[l].reshape(n * channels, -1, 1, 1) for l in range(nl)]
    biases = [parts[nl + l].reshape(n * channels) for l in range(nl)]
    return weights, biases


def _get_src_permutation_idx(indices):
    batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
    src_idx = torch.cat([src for (src, _) in indices])
    return batch_idx, src_idx


class OCPG(nn.Module):
    def __init__(self, args, backbone, transformer, num_classes, num_queries, num_feature_levels, num_frames, mask_dim,
                 dim_feedforward, controller_layers, dynamic_mask_channels, aux_loss=False, with_box_refine=False,
                 two_stage=False, freeze_text_encoder=False, freeze_video_encoder=False, rel_coord=True, matcher=None):
        super().__init__()
        assert two_stage is False, "args.two_stage must be false!"
        self.args = args
        self.matcher = matcher
        self.num_frames = num_frames
        self.num_feature_levels = num_feature_levels
        self.training = not args.eval
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.hidden_dim = hidden_dim
        self.aux_loss = aux_loss
        self.with_box_refine = with_box_refine
        self.num_classes = num_classes
        self.class_embed = amp_cache.Linear(hidden_dim, num_classes)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.ls_feat_viz = amp_cache.Conv2d(hidden_dim, 8, 3, 1, 1)
        self.ls_text_proj = amp_cache.Linear(hidden_dim, 8)
        self.mask_dim = mask_dim
        self.controller_layers = controller_layers
        self.dynamic_mask_channels = dynamic_mask_channels
        self.backbone = backbone
        if freeze_video_encoder:
            for p in self.backbone.parameters():
                p.requires_grad_(False)

        self.text_encoder = TextEncoder(args)
        self.text_pos = PositionEmbeddingSine1D(hidden_dim, normalize=True)
        self.text_proj = FeatureResizer(self.text_encoder.feat_dim, hidden_dim, dropout=0.1)
        self.sentence_proj = FeatureResizer(self.text_encoder.feat_dim, hidden_dim, dropout=0.1)
        self.fusion_module = VisionLanguageFusionModule(d_model=hidden_dim, nhead=8)

        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        if num_feature_levels <= 1:
            raise NotImplementedError
        chans = backbone.num_channels[-3:]
        proj, fft, fft_post = [], [], []
        for cin in chans:
            proj.append(nn.Sequential(amp_cache.Conv2d(cin, hidden_dim, kernel_size=1), GroupNorm(32, hidden_dim)))
            amp_cache.mark_single_use(proj[-1][0])
        cin = chans[-1]
        for _ in range(num_feature_levels - len(chans)):
            proj.append(nn.Sequential(amp_cache.Conv2d(cin, hidden_dim, kernel_size=3, stride=2, padding=1), GroupNorm(32, hidden_dim)))
            cin = hidden_dim
        for _ in range(len(proj)):
            fft.append(LFMResizeAdaptive(hidden_dim, 7))
            fft_post.append(LFMResizeAdaptive(hidden_dim, 7))
        self.input_proj = nn.ModuleList(proj)
        self.input_fft = nn.ModuleList(fft)
        self.input_fft_post = nn.ModuleList(fft_post)

        self.mask_refine = MSO(mask_dim=dynamic_mask_channels, img_dim=backbone.num_channels[:2], out_dim=dynamic_mask_channels)
        self.rel_coord = rel_coord
        self.init_aux_head()
        self.build_controller()
        self.transformer.decoder.compute_samples = False      # dead output in the reference's model (never read): not computed

    # ------------------------------------------------------------------------------------------------------
    def init_aux_head(self):
        prior = 0.01
        self.class_embed.bias.data = torch.ones(self.num_classes) * (-math.log((1 - prior) / prior))
        nn.init.zeros_(self.bbox_embed.layers[-1].weight)
        nn.init.zeros_(self.bbox_embed.layers[-1].bias)
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.zeros_(proj[0].bias)
        nn.init.xavier_uniform_(self.ls_feat_viz.weight, gain=1)
        nn.init.zeros_(self.ls_feat_viz.bias)
        n_pred = self.transformer.decoder.num_layers
        if self.with_box_refine:
            self.class_embed = _get_clones(self.class_embed, n_pred)
            self.bbox_embed = _get_clones(self.bbox_embed, n_pred)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            self.transformer.decoder.bbox_embed = self.bbox_embed          # shared module: iterative refinement
        else:
            nn.init.constant_(self.bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([self.class_embed for _ in range(n_pred)])
            self.bbox_embed = nn.ModuleList([self.bbox_embed for _ in range(n_pred)])
            self.transformer.decoder.bbox_embed = None

    def build_controller(self):
        self.in_channels = self.mask_dim
        self.mask_out_stride = 4
        self.mask_feat_stride = 8
        ch = self.dynamic_mask_channels
        self.weight_nums, self.bias_nums = [], []
        for l in range(self.controller_layers):
            cin = (self.in_channels + (2 if self.rel_coord else 0)) if l == 0 else ch
            self.weight_nums.append(cin * ch)
            self.bias_nums.append(ch)
        self.num_gen_params = sum(self.weight_nums) + sum(self.bias_nums)
        self.controller = MLP(self.hidden_dim, self.hidden_dim, self.num_gen_params, 3)
        for layer in self.controller.layers:
            nn.init.zeros_(layer.bias)
            nn.init.xavier_uniform_(layer.weight)

    # ------------------------------------------------------------------------------------------------------
    def forward_text(self, captions, device):
        if isinstance(captions, PrecomputedText) or isinstance(captions[0], str):
            feats, sentence, pad = self.text_encoder(captions, device)
            return NestedTensor(self.text_proj(feats), pad), self.sentence_proj(sentence)
        raise ValueError("Please mask sure the caption is a list of string")

    def _fuse_level(self, l, src, b, t, text_words, text_pad, text_pos, high_filter):
        """LFM -> text gating -> LFM on one level ([(b t), C, h, w])."""
        n, c, h, w = src.shape
        src, high_filter = self.input_fft[l](src, high_filter)
        if CL_FUSE and GATE_BF and src.is_cuda and src.permute(0, 2, 3, 1).is_contiguous():
            # the channels-last map IS the token list [b, (t h w), c]: gate it where it lies (no token-major round trip)
            out = self.fusion_module.forward_batch_first(src.permute(0, 2, 3, 1).reshape(b, t * h * w, c), text_words, text_pad, text_pos)
            if out is not None:
                return self.input_fft_post[l](out.view(b * t, h, w, c).permute(0, 3, 1, 2), high_filter)
        vis = src.view(b, t, c, h, w).permute(1, 3, 4, 0, 2)                      # t h w b c
        vis = self.fusion_module(visual=vis, text=text_words, text_key_padding_mask=text_pad, text_pos=text_pos, visual_pos=None)
        if CL_FUSE and vis.is_cuda:
            # [(b t), c, h, w] in channels-last memory: ONE copy out of the token-major layout (the LFM's transforms and the transformer's
            # flatten(2).transpose(1, 2) both read channels-last; NCHW here cost a second transposing copy in each of them)
            src = vis.view(t, h, w, b, c).permute(3, 0, 1, 2, 4).reshape(b * t, h, w, c).permute(0, 3, 1, 2)
        else:
            src = vis.view(t, h, w, b, c).permute(3, 0, 4, 1, 2).reshape(b * t, c, h, w)
        return self.input_fft_post[l](src, high_filter)

    def forward(self, samples, captions, targets):
        """samples: NestedTensor([B,T,3,H,W], mask [B,T,H,W]) or list of [T,3,H,W]; captions: list[str] (or
        PrecomputedText); targets: list[dict] (needs 'size'; training also needs the matcher/criterion keys)."""
        with amp_cache.scope(self):        # one fused low-precision cast of all autocast-consumed parameters
            return self._forward(samples, captions, targets)

    def _forward(self, samples, captions, targets):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_videos_list(samples, 1 if self.training else 16)
        features, visual_pos = self.backbone(samples)        # NB: folds samples to [(b t), ...] in place
        b = len(targets)
        t = visual_pos[0].shape[0] // b
        dev = visual_pos[0].device

        if "valid_indices" in targets[0]:      # A2D / JHMDB: one annotated frame per clip
            vi = torch.tensor([i * t + tg["valid_indices"] for i, tg in enumerate(targets)]).to(dev)
            for f in features:
                f.tensors, f.mask = f.tensors.index_select(0, vi), f.mask.index_select(0, vi)
            visual_pos = [p.index_select(0, vi) for p in visual_pos]
            samples.mask, samples.tensors = samples.mask.index_select(0, vi), samples.tensors.index_select(0, vi)
            t = 1

        text_features, text_sentence = self.forward_text(captions, device=dev)
        text_pos = self.text_pos(text_features).permute(2, 0, 1)                 # [L, B, C]
        text_words, text_pad = text_features.decompose()
        text_words = text_words.permute(1, 0, 2)

        # ---- spectrum-guided cross-modal fusion (last three backbone levels + one extra stride-2 level) ----
        srcs, masks, poses = [], [], []
        high_filter = None
        n_scales = 3
        for l, (feat, pos_l) in enumerate(zip(features[-n_scales:], visual_pos[-n_scales:])):
            src, mask = feat.decompose()
            assert mask is not None
            src, high_filter = self._fuse_level(l, self.input_proj[l](src), b, t, text_words, text_pad, text_pos, high_filter)
            srcs.append(src), masks.append(mask), poses.append(pos_l)
        for l in range(len(srcs), self.num_feature_levels):
            src = self.input_proj[l](features[-1].tensors if l == n_scales else srcs[-1])
            mask = resize_mask(samples.mask, src.shape[-2:])
            pos_l = self.backbone[1](NestedTensor(src, mask)).to(src.dtype)
            src, high_filter = self._fuse_level(l, src, b, t, text_words, text_pad, text_pos, high_filter)
            srcs.append(src), masks.append(mask), poses.append(pos_l)

        # ---- deformable transformer ----
        text_embed = text_sentence[:, None, None, :].expand(-1, t, self.num_queries, -1)
        hs, memory, init_reference, inter_references, box_deltas, _, inter_samples = \
            self.transformer(srcs, text_embed, masks, poses, self.query_embed.weight)
        nl = hs.shape[0]

        # ---- class / box heads ----
        out = {}
        classes, coords = [], []
        hs_l = hs.unbind(0)              # one StackBackward instead of a zeros + copy + add per level
        refs_l = (init_reference,) + tuple(inter_references.unbind(0)[:-1])
        for lvl in range(nl):
            ref = inverse_sigmoid(refs_l[lvl])
            delta = box_deltas[lvl] if box_deltas is not None else self.bbox_embed[lvl](hs_l[lvl])
            if ref.shape[-1] == 4:
                delta = delta + ref
            else:
                assert ref.shape[-1] == 2
                delta = delta + F.pad(ref, (0, 2))          # == cat([delta[..., :2] + ref, delta[..., 2:]]): no slice backward (zeros + copy + add)
            classes.append(self.class_embed[lvl](hs_l[lvl]))
            coords.append(delta.sigmoid())
        outputs_class = torch.stack(classes).unflatten(1, (b, t))                 # [l, b, t, q, k]
        outputs_coord = torch.stack(coords).unflatten(1, (b, t))                  # [l, b, t, q, 4]
        out["pred_logits"], out["pred_boxes"] = outputs_class[-1], outputs_coord[-1]

        # ---- dynamic-conv mask head ----
        tar = memory[0].shape[-2:]
        memory_fusion = sum(bicubic_resize(x.float(), tar) for x in memory)
        mask_features = memory_fusion.unflatten(0, (b, t))                         # [b, t, C, h, w]
        # all decoder layers in ONE head evaluation: the nl*q parameter sets of a frame sit next to each other ([b, t, l, q]),
        # the mask features are shared (the reference loops over the layers, ocpg.py:339-349)
        nq = self.num_queries
        params = self.controller(hs.transpose(0, 1)).reshape(b, t * nl * nq, -1)                 # hs [l, b*t, q, c]
        refs = inter_references[..., :2].transpose(0, 1).reshape(b, t * nl * nq, 2)
        m_all = self.dynamic_mask_with_coords(mask_features, params, refs, targets)             # [b, (t l q), 16, h, w]
        sh_all = F.pixel_shuffle(m_all.flatten(0, 1), 4).squeeze(1).view(b, t, nl, nq, 4 * tar[0], 4 * tar[1])
        m_all = m_all.view(b, t, nl, nq, 16, tar[0], tar[1])
        seg_masks = [m_all[:, :, l] for l in range(nl)]                                         # [b, t, q, 16, h, w] views
        seg_masks_shuffled = [sh_all[:, :, l] for l in range(nl)]                               # [b, t, q, 4h, 4w] views

        if self.training:
            # in-forward matching (ocpg.py:352-366), all decoder layers in ONE tensor program
            shuffled = sh_all.permute(2, 0, 1, 3, 4, 5)                               # [l, b, t, q, 4h, 4w] (view)
            with torch.no_grad():
                if self.aux_loss:
                    src_all = self.matcher.match_stacked(outputs_class, outputs_coord, shuffled, targets)      # [l, b]
                else:
                    src_all = self.matcher.match_stacked(outputs_class[-1:], outputs_coord[-1:], shuffled[-1:], targets)
                out["pred_masks"] = seg_masks_shuffled[-1]
                out["main_matcher_index"] = self.matcher.as_indices(src_all[-1])
                if self.aux_loss:
                    out["aux_outputs"] = self._set_aux_loss(outputs_class, outputs_coord, seg_masks_shuffled)
                    out["aux_matcher_index"] = [self.matcher.as_indices(src_all[i]) for i in range(nl - 1)]
            if self.aux_loss:
                lsf = self._ls_feat(memory_fusion)
                ls_feat = (bilinear_resize(lsf, (4 * lsf.shape[-2], 4 * lsf.shape[-1]), True) if lsf.is_cuda and MATMUL_BILINEAR
                           else F.interpolate(lsf, scale_factor=4, mode="bilinear", align_corners=True))
                ls_feat = ls_feat.unflatten(0, (b, t))                              # [b, t, 8, 4h, 4w]
                txt = self.ls_text_proj(text_sentence)[:, None, :, None, None]
                sim = (ls_feat * txt).sum(dim=2) / ((F.normalize(ls_feat, dim=2) * F.normalize(txt, dim=2)).sum(dim=2) + 1e-5)
                img = (bilinear_resize(samples.tensors, tuple(ls_feat.shape[-2:]), True) if lsf.is_cuda and MATMUL_BILINEAR
                       else F.interpolate(samples.tensors, ls_feat.shape[-2:], mode="bilinear", align_corners=True)).unflatten(0, (b, t))
                ls_features = torch.cat([img, ls_feat, sim.unsqueeze(2)], dim=2)    # [b, t, 12, 4h, 4w]  (same for every query)

                # the matched query of every layer: one gather over [l, b, t, q, ...]
                seg = m_all.permute(2, 0, 1, 3, 4, 5, 6)                            # [l, b, t, q, 16, h, w] (view)
                gi = src_all[:, :, None, None, None, None, None].expand(nl, b, t, 1, 16, tar[0], tar[1])
                picked = torch.gather(seg, 3, gi).squeeze(3).flatten(1, 2)          # [l, (b t), 16, h, w]
                refined = self.mask_refine.forward_multi(list(picked.unbind(0)), features[:2], stacked=True)   # [l*(b t), 1, 2h, 2w]
                refined = nearest_upsample(refined, 4).squeeze(1)
                refined = refined.view(nl, b, t, *refined.shape[-2:])
                gl = src_all[:, :, None, None, None, None].expand(nl, b, t, 1, 4 * tar[0], 4 * tar[1])
                low = torch.gather(shuffled, 3, gl).squeeze(3)                      # [l, b, t, 4h, 4w]
                out["pred_masks"] = refined[-1]
                out["ls_features"] = ls_features
                out["frames"] = img
                out["pred_masks_low"] = low[-1]
                out["aux_outputs"] = self._set_aux_loss_comprehensive(outputs_class, outputs_coord, refined, low, ls_features, img)
                # layer-stacked views for the criterion, in ITS call order (main, aux 0, aux 1, ...): no re-stacking copies
                def main_first(x):      # rotate the last layer to the front: one kernel each way (slices + cat: 2 x (zeros + copy) + add backward)
                    return torch.roll(x, 1, 0)
                out["_stacked"] = {"pred_logits": main_first(outputs_class), "pred_boxes": main_first(outputs_coord),
                                   "pred_masks": main_first(refined), "pred_masks_low": main_first(low)}
        elif self.args.dataset_file not in ("a2d", "jhmdb") and "refcoco" not in self.args.dataset_file:
            # YTVOS / DAVIS: keep the clip's best query (mean sigmoid score over frames), refine only that one
            m = seg_masks[-1].view(b, t, self.num_queries, 16, tar[0], tar[1])
            scores = out["pred_logits"].sigmoid().mean(1).max(-1)[0]               # [b, q]
            best = scores.argmax(-1)                                                # [b]
            bi = torch.arange(b, device=dev)
            out["pred_logits"] = out["pred_logits"][bi, :, best][:, :, None]
            out["pred_boxes"] = out["pred_boxes"][bi, :, best][:, :, None]
            out["reference_points"] = inter_references[-2].unflatten(0, (b, t))[bi, :, best][:, :, None, :2]
            pm = self.mask_refine(m[bi, :, best].flatten(0, 1), features[:2])
            pm = F.interpolate(pm, scale_factor=4).squeeze(1).unflatten(0, (b, t))
            out["pred_masks"] = pm.unsqueeze(2)
        else:
            m = seg_masks[-1].view(b, t, self.num_queries, 16, tar[0], tar[1])
            per_q = []
            for qi in range(self.num_queries):
                pm = self.mask_refine(m[:, :, qi].flatten(0, 1), features[:2])
                per_q.append(F.interpolate(pm, scale_factor=4).squeeze(1).unflatten(0, (b, t)))
            out["pred_masks"] = torch.stack(per_q, dim=2)
        return out

    @torch.jit.unused
    def _ls_feat(self, x):
        """ls_feat_viz (ocpg.py:71 of the reference: Conv2d(hidden, 8, 3, 1, 1)) on the GPU: the <= 16-output-channel MFMA kernels of
        csrc/mso.hip (forward, input gradient, weight + bias gradient) on the channels-last map, operands in the autocast dtype with fp32
        accumulation as an autocast convolution has; the library convolution otherwise."""
        conv = self.ls_feat_viz
        if not (LS_FEAT_N16 and x.is_cuda and x.shape[0] <= 65535 and x.dtype in (torch.float32, torch.bfloat16, torch.float16)):
            return conv(x)
        from .decoder import _bias32, _nhwc, _tap_major
        from .ops.functions.mso_func import compute_code, conv3x3_n16
        cdt = compute_code(x.device.type)
        y = conv3x3_n16(_nhwc(x), _tap_major(amp_cache.lookup(conv.weight), cdt), _bias32(conv.bias), cdt=cdt)       # [N, h, w, 8] fp32
        y = y.permute(0, 3, 1, 2)
        return y if cdt == 0 else y.to(torch.get_autocast_dtype(x.device.type))          # an autocast convolution returns the autocast dtype

    def _set_aux_loss(self, outputs_class, outputs_coord, outputs_seg_masks):
        return [{"pred_logits": a, "pred_boxes": b_, "pred_masks": c}
                for a, b_, c in zip(outputs_class[:-1], outputs_coord[:-1], outputs_seg_masks[:-1])]

    @torch.jit.unused
    def _set_aux_loss_comprehensive(self, outputs_class, outputs_coord, seg_masks, seg_masks_low, ls_features, img_ori):
        return [{"pred_logits": a, "pred_boxes": b_, "pred_masks": c, "pred_masks_low": d, "ls_features": ls_features, "frames": img_ori}
                for a, b_, c, d in zip(outputs_class[:-1], outputs_coord[:-1], seg_masks[:-1], seg_masks_low[:-1])]

    # ------------------------------------------------------------------------------------------------------
    def dynamic_mask_with_coords(self, mask_features, mask_head_params, reference_points, targets):
        """Per-query dynamic 1x1 convs (C[+2] -> 16 -> 16 ...) over the stride-8 mask features.

        mask_features [b,t,C,h,w]; mask_head_params [b, t*q, n_params]; reference_points [b, t*q, 2] (cx, cy in [0,1]);
        targets[i]['size'] = (img_h, img_w).  Returns [b, t*q, 16, h, w].
        The two relative-coordinate input channels are (ref_x*img_w - x_pix, ref_y*img_h - y_pix) in raw input
        pixels with x_pix = 8*col + 4 (ocpg.py:496-511) and are handled in closed form, fp32 always.
        """
        with torch.autocast(device_type=mask_features.device.type, enabled=False):
            feats = mask_features.float()
            b, t, c, h, w = feats.shape
            nq = reference_points.shape[1] // t
            ch = self.dynamic_mask_channels
            if feats.is_cuda and self.rel_coord and self.controller_layers == 2 and ch == 16:
                # fused HIP forward (csrc/dynmask.hip): coordinates + both per-query 1x1 convs in one pass over the features
                sizes = torch.stack([tg["size"] for tg in targets]).to(feats.device, torch.float32)         # [b, 2] (h, w)
                refpix = reference_points.float() * torch.stack([sizes[:, 1], sizes[:, 0]], -1)[:, None, :]  # [b, t*q, 2]
                out = dynamic_mask(feats.reshape(b * t, c, h, w), mask_head_params.float().reshape(b * t * nq, -1),
                                   refpix.reshape(b * t * nq, 2), self.mask_feat_stride)
                return out.view(b, t * nq, ch, h, w)
            params = mask_head_params.float().reshape(b, t, nq, -1)
            parts = torch.split_with_sizes(params, self.weight_nums + self.bias_nums, dim=-1)
            n_layers = len(self.weight_nums)
            weights, biases = parts[:n_layers], parts[n_layers:]
            cin0 = c + (2 if self.rel_coord else 0)
            w0 = weights[0].reshape(b, t, nq * ch, cin0)
            x = torch.matmul(w0[..., :c].reshape(b * t, nq * ch, c), feats.reshape(b * t, c, h * w))        # [bt, q*16, hw]
            x = x.view(b, t, nq, ch, h, w)
            if self.rel_coord:
                sizes = torch.stack([tg["size"] for tg in targets]).to(feats.device, torch.float32)         # [b, 2] (h, w)
                ref = reference_points.float().reshape(b, t, nq, 2) * torch.stack([sizes[:, 1], sizes[:, 0]], -1)[:, None, None, :]
                s = self.mask_feat_stride
                xs = torch.arange(0, w * s, step=s, dtype=torch.float32, device=feats.device) + s // 2
                ys = torch.arange(0, h * s, step=s, dtype=torch.float32, device=feats.device) + s // 2
                wx = w0[..., c].reshape(b, t, nq, ch)
                wy = w0[..., c + 1].reshape(b, t, nq, ch)
                relx = ref[..., 0, None] - xs                                           # [b,t,q,w]
                rely = ref[..., 1, None] - ys                                           # [b,t,q,h]
                x = x + wx[..., None, None] * relx[:, :, :, None, None, :] + wy[..., None, None] * rely[:, :, :, None, :, None]
            x = x + biases[0].reshape(b, t, nq, ch, 1, 1)
            for l in range(1, n_layers):
                x = F.relu(x)
                wl = weights[l].reshape(b * t * nq, ch, ch)
                x = torch.bmm(wl, x.reshape(b * t * nq, ch, h * w)).view(b, t, nq, ch, h, w) + biases[l].reshape(b, t, nq, ch, 1, 1)
            return x.reshape(b, t * nq, ch, h, w)


def build(args):
    if args.binary:
        num_classes = 1
    else:
        num_classes = {"ytvos": 65, "davis": 78, "a2d": 1, "jhmdb": 1}.get(args.dataset_file, 91)
    device = torch.device(args.device)
    if "video_swin" in args.backbone:
        from .video_swin_transformer import build_video_swin_backbone
        backbone = build_video_swin_backbone(args)
    elif "swin" in args.backbone:
        raise NotImplementedError("2-D Swin image backbones are out of scope (SURVEY.md section 2.1 #8)")
    else:
        backbone = build_backbone(args)
    transformer = build_deforamble_transformer(args)
    matcher = build_matcher(args)
    model = OCPG(args, backbone, transformer, num_classes=num_classes, num_queries=args.num_queries,
                 num_feature_levels=args.num_feature_levels, num_frames=args.num_frames, mask_dim=args.mask_dim,
                 dim_feedforward=args.dim_feedforward, controller_layers=args.controller_layers,
                 dynamic_mask_channels=args.dynamic_mask_channels, aux_loss=args.aux_loss,
                 with_box_refine=args.with_box_refine, two_stage=args.two_stage,
                 freeze_text_encoder=args.freeze_text_encoder, freeze_video_encoder=args.freeze_video_encoder,
                 rel_coord=args.rel_coord, matcher=matcher)
    weight_dict = {"loss_ce": args.cls_loss_coef, "loss_bbox": args.bbox_loss_coef, "loss_giou": args.giou_loss_coef}
    if args.masks:
        for suffix in ("", "_low"):
            weight_dict["loss_mask" + suffix] = args.mask_loss_coef
            weight_dict["loss_dice" + suffix] = args.dice_loss_coef
            weight_dict["loss_proj" + suffix] = args.proj_loss_coef
            weight_dict["loss_lst" + suffix] = args.lst_loss_coef
    if args.aux_loss:
        base = dict(weight_dict)
        for i in range(args.dec_layers - 1):
            weight_dict.update({f"{k}_{i}": v for k, v in base.items()})
    losses = ["labels", "boxes"] + (["masks"] if args.masks else [])
    criterion = SetCriterion(args, num_classes, matcher=matcher, weight_dict=weight_dict, eos_coef=args.eos_coef,
                             losses=losses, focal_alpha=args.focal_alpha)
    criterion.to(device)
    return model, criterion, build_postprocessors(args, args.dataset_file)
