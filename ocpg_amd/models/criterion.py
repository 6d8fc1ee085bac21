"""SetCriterion -- the weakly-supervised loss of OCPG: focal classification, L1 + GIoU boxes, and for the masks
(full-res and low-res) the heat-map-weighted BCE, the box-projection loss and the level-set loss, for the last
decoder layer and every auxiliary layer (reference models/criterion.py:14-254, loss functions
models/segmentation.py:134-315).

Quirks kept on purpose: `self.iter` advances once per loss_masks call (4x per step),
the (1-warmup)/warmup blending of loss_mask / loss_lst, the low-res targets being the [1::2] sub-sampling of the
/32-padded GT, the last (similarity) channel of ls_features dropped for the level-set term, `loss_dice*` weights that
no loss ever fills.

MI355X-first structure: the reference evaluates the 9 losses once per decoder layer in a Python loop (4 x ~250 tiny
kernels forward, as many again backward).  Here all layers are evaluated TOGETHER on tensors with a leading layer axis
([Lr, B, T, ...]); everything that depends only on the targets (padding, sub-sampling, box rasterisation, heat-map
weights) is computed once.  Host-sync-free: num_boxes stays a device tensor (one all-reduce, no .item()).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..util import box_ops
from ..util.misc import get_world_size, is_dist_avail_and_initialized, nested_tensor_from_tensor_list
from .matcher import _assert_well_formed, _pairwise_giou_1
from .ops.functions import mask_loss_func
from .segmentation import generate_box_region_mask

HIP_MASK_LOSSES = True      # A/B switch: fused HIP level-set / projection losses on the GPU
HIP_DET_LOSSES = True       # A/B switch: fused HIP classification / L1 / GIoU losses on the GPU


def _pad_stack(targets, key):
    return nested_tensor_from_tensor_list([t[key] for t in targets], size_divisibility=32, split=False).decompose()[0]


def _sub(x, k):
    return x[..., k // 2::k, k // 2::k]


def _dice(x, t):
    """dice_coefficient (segmentation.py:203-211) over the trailing axis: x [..., n], t broadcastable -> [...]."""
    inter = (x * t).sum(-1)
    union = (x ** 2.0).sum(-1) + (t ** 2.0).sum(-1) + 1e-5
    return 1.0 - (2 * inter / union)


class SetCriterion(nn.Module):
    def __init__(self, args, num_classes, matcher, weight_dict, eos_coef, losses, focal_alpha=0.25):
        super().__init__()
        self.args = args
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.eos_coef = eos_coef
        self.losses = losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.focal_alpha = focal_alpha
        self.mask_out_stride = 1
        self.mask_out_stride_low = 2
        self.iter = 0
        self._warmup_iters = 100000
        # optional device-resident copy of `iter` (0-dim float tensor) for callers that replay a captured HIP graph: a
        # Python counter would be frozen into the graph; the caller advances it by the number of layers per replay
        self.iter_device = None

    # -- layer-stacked losses: every input carries a leading layer axis Lr ---------------------------------------
    def _labels_stacked(self, logits, src, targets, num_boxes):
        """logits [Lr,B,T,q,K], src [Lr,B] -> loss_ce [Lr] (criterion.py:46-83)."""
        lr, b, nf, nq, k = logits.shape
        valid = torch.stack([t["valid"] for t in targets]).to(logits.device) > 0              # [B, T]
        hit = (torch.arange(nq, device=logits.device).view(1, 1, 1, nq) == src[:, :, None, None]) & valid[None, :, :, None]
        if self.num_classes == 1:
            onehot = hit[..., None].to(logits.dtype)
        else:
            labels = torch.stack([t["labels"] for t in targets]).to(logits.device)             # [B, T]
            onehot = (F.one_hot(labels, k)[None, :, :, None, :].bool() & hit[..., None]).to(logits.dtype)
        x, t = logits.reshape(lr, b, nf * nq, k), onehot.reshape(lr, b, nf * nq, k)
        p = x.sigmoid()
        ce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
        p_t = p * t + (1 - p) * (1 - t)
        loss = ce * ((1 - p_t) ** 2)
        if self.focal_alpha >= 0:
            loss = (self.focal_alpha * t + (1 - self.focal_alpha) * (1 - t)) * loss
        return loss.mean(2).sum((1, 2)) / num_boxes * (nf * nq)

    def _boxes_stacked(self, boxes, src, targets, num_boxes):
        """boxes [Lr,B,T,q,4], src [Lr,B] -> (loss_bbox [Lr], loss_giou [Lr]) (criterion.py:85-107)."""
        lr, b, nf, nq, _ = boxes.shape
        idx = src[:, :, None, None, None].expand(lr, b, nf, 1, 4)
        sel = torch.gather(boxes, 3, idx)[:, :, :, 0]                                           # [Lr, B, T, 4]
        tgt = torch.stack([t["boxes"] for t in targets]).to(sel.dtype)[None]                    # [1, B, T, 4]
        l1 = (sel - tgt).abs().sum((1, 2, 3)) / num_boxes
        sx, tx = box_ops.box_cxcywh_to_xyxy(sel), box_ops.box_cxcywh_to_xyxy(tgt)
        _assert_well_formed(sx, "predictions")
        _assert_well_formed(tx, "targets")
        giou = (1 - _pairwise_giou_1(sx, tx)).sum((1, 2)) / num_boxes
        return l1, giou

    def _masks_stacked(self, pm, pml, src_lst, targets, num_boxes, warm):
        """pm [Lr,B,T,H,W] full-res logits, pml [Lr,B,T,h,w] low-res, src_lst [B,T,C,h,w], warm [Lr] (criterion.py:109-190).
        Always evaluated in fp32 (the reference's losses are elementwise/reduction ops that autocast leaves in fp32)."""
        with torch.autocast(device_type=pm.device.type, enabled=False):
            return self._masks_stacked_fp32(pm.float(), pml.float(), src_lst.float(), targets, num_boxes, warm)

    def _masks_stacked_fp32(self, pm, pml, src_lst, targets, num_boxes, warm):
        lr = pm.shape[0]
        gt = _pad_stack(targets, "masks").to(pml)
        heat = _pad_stack(targets, "weights").to(pml)
        weak = _pad_stack(targets, "weak_masks").to(pml)
        s, sl = self.mask_out_stride, self.mask_out_stride_low
        im_h, im_w = gt.shape[-2:]
        b, nf = weak.shape[:2]
        gt_full = _sub(gt, s)
        assert gt_full.size(2) * s == im_h and gt_full.size(3) * s == im_w
        sizes = torch.stack([t["size"] for t in targets]).repeat_interleave(nf, dim=0)
        boxes = box_ops.box_cxcywh_to_xyxy(torch.cat([t["boxes"] for t in targets], dim=0))
        region = generate_box_region_mask(boxes, (im_h, im_w), sizes).view(b, nf, im_h, im_w)
        region_low, region = _sub(region, sl), _sub(region, s)
        weak_full, weak_low = _sub(weak, s) * region, _sub(weak, sl) * region_low

        def heat_weight(h, reg):        # masked_ce_loss's weight map (segmentation.py:177-190): targets only
            w = (h.clamp(min=0.3, max=0.7) - 0.5).abs()
            lo, hi = w.min(), w.max()
            w = (w - lo) / (hi - lo + 1e-5)
            return torch.where(reg == 0, torch.ones_like(w), w)

        def masked_ce(x, w, m):         # BCE-with-logits applied to sigmoid(x)*w (the reference's double squashing)
            return F.binary_cross_entropy_with_logits(x.sigmoid() * w, (m * w).expand_as(x), reduction="none").mean((1, 2, 3, 4))

        def proj(x, reg, m):            # proj_loss with_mean_term (segmentation.py:253-277); axis 3 = rows, 4 = columns
            p = x.sigmoid()
            ly = _dice(p.amax(3).flatten(2), reg.amax(2).flatten(1)[None])
            lx = _dice(p.amax(4).flatten(2), reg.amax(3).flatten(1)[None])
            my = _dice(p.mean(3).flatten(2), m.mean(2).flatten(1)[None])
            mx = _dice(p.mean(4).flatten(2), m.mean(3).flatten(1)[None])
            return (ly + lx).mean(1) + 0.1 * (my + mx).mean(1)

        def levelset(x, feats, box):    # levelset_loss (segmentation.py:279-315): x [Lr,N,h,w], feats [N,C,h,w], box [N,h,w]
            p = x.sigmoid()
            fg, bg = p * box, (1.0 - p) * box                                                  # [Lr,N,h,w]
            pixels = box.sum((1, 2)).clamp(min=1)                                              # [N]
            tgt = feats * box[:, None]                                                          # [N,C,h,w]
            c_in = torch.einsum("lnhw,nchw->lnc", fg, tgt) / fg.sum((2, 3)).clamp(min=0.00001)[..., None]
            c_out = torch.einsum("lnhw,nchw->lnc", bg, tgt) / bg.sum((2, 3)).clamp(min=0.00001)[..., None]
            # sum_c sum_hw (tgt - c)^2 * w  =  sum_hw w * sum_c tgt^2  - 2 c . sum_hw(w tgt) + |c|^2 sum_hw w   (no [Lr,N,C,h,w] temp)
            t2 = (tgt ** 2).sum(1)                                                              # [N,h,w]
            def energy(wgt, c):
                return ((wgt * t2).sum((2, 3)) - 2 * (c * torch.einsum("lnhw,nchw->lnc", wgt, tgt)).sum(-1)
                        + (c ** 2).sum(-1) * wgt.sum((2, 3)))
            region_e = (energy(fg, c_in) + energy(bg, c_out)) / tgt.shape[1]
            def length(v):
                return (v[..., 1:, :] - v[..., :-1, :]).abs().sum((2, 3)) + (v[..., :, 1:] - v[..., :, :-1]).abs().sum((2, 3))
            return (region_e / pixels + 0.00001 * (length(fg) + length(bg)) / pixels).mean(1)

        w_full, w_low = heat_weight(_sub(heat, s), region), heat_weight(_sub(heat, sl), region_low)
        if pm.is_cuda and HIP_MASK_LOSSES and pm[0].numel() % 4 == 0 and pml[0].numel() % 4 == 0:
            loss_mask, loss_mask_low = mask_loss_func.masked_ce(pm, w_full, weak_full), mask_loss_func.masked_ce(pml, w_low, weak_low)
        else:
            loss_mask, loss_mask_low = masked_ce(pm, w_full, weak_full), masked_ce(pml, w_low, weak_low)
        lst_hw = src_lst.shape[-2:]
        scaled = F.interpolate(pm.flatten(0, 1), lst_hw, mode="bilinear", align_corners=True).view(lr, b * nf, *lst_hw)
        region_scaled = F.interpolate(region, lst_hw, mode="nearest").flatten(0, 1)
        feats = src_lst.flatten(0, 1)[:, :-1]
        if pm.is_cuda and HIP_MASK_LOSSES and feats.shape[1] <= 16:
            # fused HIP kernels (csrc/levelset.hip, csrc/proj.hip): 2-3 launches per loss instead of ~30-60 (+ twice that backward)
            full_feats = src_lst.flatten(0, 1)                     # the kernel skips the dropped last channel itself
            ls = mask_loss_func.levelset_loss(scaled, full_feats, region_scaled, feats.shape[1])
            ls_low = mask_loss_func.levelset_loss(pml.flatten(1, 2), full_feats, region_scaled, feats.shape[1])
            pj, pj_low = mask_loss_func.proj_loss(pm, region, weak_full), mask_loss_func.proj_loss(pml, region_low, weak_low)
        else:
            ls = levelset(scaled, feats, region_scaled)
            ls_low = levelset(pml.flatten(1, 2), feats, region_scaled)
            pj, pj_low = proj(pm, region, weak_full), proj(pml, region_low, weak_low)
        out = {
            "loss_proj": pj, "loss_mask": (1 - warm) * loss_mask, "loss_lst": warm * ls,
            "loss_proj_low": pj_low, "loss_mask_low": (1 - warm) * loss_mask_low, "loss_lst_low": warm * ls_low,
        }
        return out, (pm[0].sigmoid(), gt_full, weak_full)

    # -- reference-shaped single-layer entry points (models/criterion.py:46-211) ---------------------------------
    @staticmethod
    def _src_index(indices):
        return torch.cat([src for (src, _) in indices])

    def _get_src_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, self._src_index(indices)

    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        v = self._labels_stacked(outputs["pred_logits"][None], self._src_index(indices)[None], targets, num_boxes)
        return {"loss_ce": v[0]}, None, None, None

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        l1, giou = self._boxes_stacked(outputs["pred_boxes"][None], self._src_index(indices)[None], targets, num_boxes)
        return {"loss_bbox": l1[0], "loss_giou": giou[0]}, None, None, None

    def loss_masks(self, outputs, targets, indices, num_boxes):
        d, maps = self._masks_stacked(outputs["pred_masks"][None], outputs["pred_masks_low"][None], outputs["ls_features"],
                                      targets, num_boxes, self._warm(1).to(outputs["pred_masks"].device))
        return {k: v[0] for k, v in d.items()}, maps[0], maps[1], maps[2]

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        table = {"labels": self.loss_labels, "boxes": self.loss_boxes, "masks": self.loss_masks}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, indices, num_boxes, **kwargs)

    def _warm(self, n):
        """Warm-up factors of the next n loss_masks calls ([n] tensor or list of floats) and advance the call counter."""
        if self.iter_device is not None:
            k = torch.arange(1, n + 1, device=self.iter_device.device, dtype=torch.float32)
            return torch.clamp((self.iter_device + k) / float(self._warmup_iters), max=1.0)
        w = [min(float(self.iter + i) / float(self._warmup_iters), 1.0) for i in range(1, n + 1)]
        self.iter += n
        return torch.tensor(w)

    @staticmethod
    def global_num_boxes(targets, device):
        """Valid target frames averaged over ranks, clamped to >= 1 (criterion.py:224-231) -- a 0-dim device tensor; one
        all-reduce, no .item().  Exposed so a caller that replays a captured HIP graph can run the collective outside it."""
        num_boxes = torch.stack([t["valid"] for t in targets]).sum().to(device=device, dtype=torch.float).reshape(1)
        if is_dist_avail_and_initialized():
            torch.distributed.all_reduce(num_boxes)
        return torch.clamp(num_boxes / get_world_size(), min=1)[0]

    def forward(self, outputs, targets):
        """-> (losses dict, src_map, tgt_map, weak_map).  Layer order of the reference's calls: main, aux 0, aux 1, ..."""
        aux = outputs.get("aux_outputs", [])
        aux_indices = outputs["aux_matcher_index"] if aux else []
        assert len(aux_indices) == len(aux), "Aux index len not match."
        layers = [outputs] + list(aux)
        src = torch.stack([self._src_index(i) for i in [outputs["main_matcher_index"]] + list(aux_indices)])   # [Lr, B]
        num_boxes = outputs.get("num_boxes")       # optional: precomputed by the caller (see global_num_boxes)
        if num_boxes is None:
            num_boxes = self.global_num_boxes(targets, outputs["pred_logits"].device)
        pre = outputs.get("_stacked")              # the model hands over layer-stacked tensors (no re-stacking copies)

        def stacked(key):
            if pre is not None and key in pre:
                return pre[key]
            return torch.stack([l[key] for l in layers])

        losses, maps = {}, (None, None, None)
        per_layer = {}
        if HIP_DET_LOSSES and "labels" in self.losses and "boxes" in self.losses and src.is_cuda:
            # one launch (csrc/det_loss.hip) instead of ~70 tiny kernels forward and ~140 backward
            from .matcher import _BOX_ERRORS
            dev = src.device
            flag = _BOX_ERRORS.get(dev)
            if flag is None and not torch.cuda.is_current_stream_capturing():
                flag = _BOX_ERRORS[dev] = torch.zeros((), dtype=torch.int32, device=dev)
            valid = torch.stack([t["valid"] for t in targets]).to(dev)
            labels = None if self.num_classes == 1 else torch.stack([t["labels"] for t in targets]).to(dev)
            tboxes = torch.stack([t["boxes"] for t in targets]).to(dev)
            det = mask_loss_func.det_losses(stacked("pred_logits"), stacked("pred_boxes"), src, valid, labels, tboxes, num_boxes,
                                            self.focal_alpha, flag)
            per_layer["loss_ce"], per_layer["loss_bbox"], per_layer["loss_giou"] = det.unbind(0)
        else:
            if "labels" in self.losses:
                per_layer["loss_ce"] = self._labels_stacked(stacked("pred_logits"), src, targets, num_boxes)
            if "boxes" in self.losses:
                per_layer["loss_bbox"], per_layer["loss_giou"] = self._boxes_stacked(stacked("pred_boxes"), src, targets, num_boxes)
        if "masks" in self.losses:
            warm = self._warm(len(layers)).to(outputs["pred_masks_low"].device)
            d, maps = self._masks_stacked(stacked("pred_masks"), stacked("pred_masks_low"), outputs["ls_features"], targets, num_boxes, warm)
            per_layer.update(d)
        order = [k for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_proj", "loss_mask", "loss_lst", "loss_proj_low", "loss_mask_low",
                             "loss_lst_low") if k in per_layer]
        table = torch.stack([per_layer[k] for k in order])                     # [n_losses, Lr]
        names = [[k if i == 0 else f"{k}_{i - 1}" for i in range(len(layers))] for k in order]
        for i in range(len(layers)):            # key order of the reference: the main layer's losses, then aux 0, aux 1, ...
            for j, row in enumerate(names):
                losses[row[i]] = table[j, i]
        # the weighted total the training loop forms next (engine.py:56), as ONE reduction over the table instead of 36
        # scalar multiplies and adds (and, in backward, 36 select-scatter nodes)
        self._last = (losses, table, names)
        return (losses,) + maps

    def weighted_sum(self, loss_dict):
        """sum(loss_dict[k] * weight_dict[k] for k in loss_dict if k in weight_dict), the reference training loop's total
        (engine.py:55-56); for the dict this criterion just returned it is one multiply + one sum over the loss table."""
        last = getattr(self, "_last", None)
        if last is None or last[0] is not loss_dict:
            return sum(loss_dict[k] * self.weight_dict[k] for k in loss_dict if k in self.weight_dict)
        _, table, names = last
        self._last = None       # consumed: a graph kept alive here would hand its AccumulateGrad nodes (and their stream) to the next step
        key = (tuple(map(tuple, names)), str(table.device))
        cache = self.__dict__.setdefault("_weight_tables", {})
        w = cache.get(key)
        if w is None:       # weight_dict is fixed at construction (models/ocpg.py:573-594)
            w = cache[key] = torch.tensor([[float(self.weight_dict.get(n, 0.0)) for n in row] for row in names], dtype=torch.float32,
                                          device=table.device)
        return (table.float() * w).sum()
