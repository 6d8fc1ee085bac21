"""SetCriterion -- the weakly-supervised loss of OCPG: focal classification, L1 + GIoU boxes, and for the masks
(full-res and low-res) the heat-map-weighted BCE, the box-projection loss and the level-set loss, for the last
decoder layer and every auxiliary layer (reference models/criterion.py:14-254).

Quirks kept on purpose: `self.iter` advances once per loss_masks call (4x per step),
the (1-warmup)/warmup blending of loss_mask / loss_lst, the low-res targets being the [1::2] sub-sampling of the
/32-padded GT, the last (similarity) channel of ls_features dropped for the level-set term.
Host-sync-free: num_boxes stays a device tensor (one all-reduce, no .item()), matched-query gathers are index
arithmetic, box regions are rasterised without Python loops.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..util import box_ops
from ..util.misc import get_world_size, is_dist_avail_and_initialized, nested_tensor_from_tensor_list
from .matcher import _assert_well_formed, _pairwise_giou_1
from .segmentation import generate_box_region_mask, levelset_loss, masked_ce_loss, proj_loss, sigmoid_focal_loss


def _pad_stack(targets, key):
    return nested_tensor_from_tensor_list([t[key] for t in targets], size_divisibility=32, split=False).decompose()[0]


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
        # Python counter would be frozen into the graph; the caller advances it by 4 (calls per forward) per replay
        self.iter_device = None
        self._calls = 0

    # -- helpers ------------------------------------------------------------------------------------------
    @staticmethod
    def _src_index(indices):
        """[B] int64 matched query per clip."""
        return torch.cat([src for (src, _) in indices])

    def _get_src_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, self._src_index(indices)

    # -- losses -------------------------------------------------------------------------------------------
    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        logits = outputs["pred_logits"]                                  # [B, T, q, K]
        b, nf, nq, k = logits.shape
        src = self._src_index(indices)                                   # [B]
        valid = torch.stack([t["valid"] for t in targets]).to(logits.device) > 0          # [B, T]
        hit = (torch.arange(nq, device=logits.device)[None, None, :] == src[:, None, None]) & valid[:, :, None]
        if self.num_classes == 1:
            onehot = hit[..., None].to(logits.dtype)
        else:
            labels = torch.stack([t["labels"] for t in targets]).to(logits.device)         # [B, T]
            onehot = (F.one_hot(labels, k)[:, :, None, :].bool() & hit[..., None]).to(logits.dtype)
        loss = sigmoid_focal_loss(logits.reshape(b, nf * nq, k), onehot.reshape(b, nf * nq, k), num_boxes,
                                  alpha=self.focal_alpha, gamma=2) * (nf * nq)
        return {"loss_ce": loss}, None, None, None

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        boxes = outputs["pred_boxes"]                                    # [B, T, q, 4]
        b, nf, nq, _ = boxes.shape
        src = self._src_index(indices)
        sel = boxes[torch.arange(b, device=boxes.device), :, src].reshape(b * nf, 4)
        tgt = torch.cat([t["boxes"] for t in targets], dim=0).to(sel.dtype)
        l1 = (sel - tgt).abs().sum() / num_boxes
        sx, tx = box_ops.box_cxcywh_to_xyxy(sel), box_ops.box_cxcywh_to_xyxy(tgt)
        _assert_well_formed(sx, "predictions")
        _assert_well_formed(tx, "targets")
        giou = (1 - _pairwise_giou_1(sx, tx)).sum() / num_boxes
        return {"loss_bbox": l1, "loss_giou": giou}, None, None, None

    def loss_masks(self, outputs, targets, indices, num_boxes):
        src_masks, src_low, src_lst = outputs["pred_masks"], outputs["pred_masks_low"], outputs["ls_features"]
        gt = _pad_stack(targets, "masks").to(src_low)
        heat = _pad_stack(targets, "weights").to(src_low)
        weak = _pad_stack(targets, "weak_masks").to(src_low)
        s, sl = self.mask_out_stride, self.mask_out_stride_low
        im_h, im_w = gt.shape[-2:]
        nf = weak.shape[1]

        def sub(x, k):
            return x[:, :, k // 2::k, k // 2::k]

        gt_full = sub(gt, s)
        assert gt_full.size(2) * s == im_h and gt_full.size(3) * s == im_w
        self.iter += 1
        sizes = torch.stack([t["size"] for t in targets]).repeat_interleave(nf, dim=0)
        boxes = box_ops.box_cxcywh_to_xyxy(torch.cat([t["boxes"] for t in targets], dim=0))
        region = generate_box_region_mask(boxes, (im_h, im_w), sizes).view(-1, nf, im_h, im_w)
        region_low, region = sub(region, sl), sub(region, s)
        if self.iter_device is None:
            warm = min(float(self.iter) / float(self._warmup_iters), 1.0)
        else:   # graph-replay mode: the call counter lives on the device; `_calls` = position of this call inside forward
            self._calls += 1
            warm = torch.clamp((self.iter_device + self._calls) / float(self._warmup_iters), max=1.0)
        weak_full, weak_low = sub(weak, s) * region, sub(weak, sl) * region_low

        loss_mask, _ = masked_ce_loss(src_masks, sub(heat, s), weak_full, region, num_boxes)
        loss_mask_low, _ = masked_ce_loss(src_low, sub(heat, sl), weak_low, region_low, num_boxes)

        lst_hw = src_lst.shape[-2:]
        scaled = F.interpolate(src_masks, lst_hw, mode="bilinear", align_corners=True)
        region_scaled = F.interpolate(region, lst_hw, mode="nearest")
        feats = src_lst.flatten(0, 1)[:, :-1]
        ls = levelset_loss(scaled.flatten(0, 1)[:, None], feats, region_scaled.flatten(0, 1)[:, None])
        ls_low = levelset_loss(src_low.flatten(0, 1)[:, None], feats, region_scaled.flatten(0, 1)[:, None])
        losses = {
            "loss_proj": proj_loss(src_masks, region, weak_full, num_boxes, with_mean_term=True),
            "loss_mask": (1 - warm) * loss_mask,
            "loss_lst": warm * ls,
            "loss_proj_low": proj_loss(src_low, region_low, weak_low, num_boxes, with_mean_term=True),
            "loss_mask_low": (1 - warm) * loss_mask_low,
            "loss_lst_low": warm * ls_low,
        }
        return losses, src_masks.sigmoid(), gt_full, weak_full

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        table = {"labels": self.loss_labels, "boxes": self.loss_boxes, "masks": self.loss_masks}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, indices, num_boxes, **kwargs)

    @staticmethod
    def global_num_boxes(targets, device):
        """Valid target frames averaged over ranks, clamped to >= 1 (criterion.py:224-231) -- a 0-dim device tensor; one
        all-reduce, no .item().  Exposed so a caller that replays a captured HIP graph can run the collective outside it."""
        num_boxes = torch.stack([t["valid"] for t in targets]).sum().to(device=device, dtype=torch.float).reshape(1)
        if is_dist_avail_and_initialized():
            torch.distributed.all_reduce(num_boxes)
        return torch.clamp(num_boxes / get_world_size(), min=1)[0]

    def forward(self, outputs, targets):
        self._calls = 0
        indices = outputs["main_matcher_index"]
        aux_indices = outputs["aux_matcher_index"]
        num_boxes = outputs.get("num_boxes")       # optional: precomputed by the caller (see global_num_boxes)
        if num_boxes is None:
            num_boxes = self.global_num_boxes(targets, outputs["pred_masks_low"].device)

        losses, maps = {}, (None, None, None)
        for loss in self.losses:
            d, src_map, tgt_map, weak_map = self.get_loss(loss, outputs, targets, indices, num_boxes)
            losses.update(d)
            if src_map is not None:
                maps = (src_map, tgt_map, weak_map)
        if "aux_outputs" in outputs:
            assert len(aux_indices) == len(outputs["aux_outputs"]), "Aux index len not match."
            for i, aux in enumerate(outputs["aux_outputs"]):
                for loss in self.losses:
                    kw = {"log": False} if loss == "labels" else {}
                    d = self.get_loss(loss, aux, targets, aux_indices[i], num_boxes, **kw)[0]
                    losses.update({f"{k}_{i}": v for k, v in d.items()})
        return (losses,) + maps
