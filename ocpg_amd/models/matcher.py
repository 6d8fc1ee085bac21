"""Query <-> target assignment.  With one referred object per clip the "Hungarian" problem is an argmin over the
[q,1] cost (reference models/matcher.py:74-171; linear_sum_assignment is imported there but never called).

cost = 2*focal-class + 5*L1 + 2*(-GIoU) + 2*mask-focal + 5*(-dice) (weights from opts.py:88-99); class cost is
averaged over the VALID frames, box costs over all frames, mask costs over all pixels of the clip.
The whole batch is one sync-free tensor program (the reference loops over B and T in Python and syncs on
`tgt_valid[t] == 0`); results are the int64 `src_ind` per clip, identical to the reference's.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..util.box_ops import box_cxcywh_to_xyxy
from ..util.misc import nested_tensor_from_tensor_list


def _pairwise_giou_1(boxes, tgt):
    """GIoU of boxes [..., 4] against one target box per leading index, tgt [..., 4] (xyxy), with the reference's
    +1e-6 smoothing (util/box_ops.py:45-85)."""
    ax0, ay0, ax1, ay1 = boxes.unbind(-1)          # one unbind (backward: one stack) instead of 12 select/slice nodes
    bx0, by0, bx1, by1 = tgt.unbind(-1)
    area_a = (ax1 - ax0) * (ay1 - ay0)
    area_b = (bx1 - bx0) * (by1 - by0)
    inter = (torch.min(ax1, bx1) - torch.max(ax0, bx0)).clamp(min=0) * (torch.min(ay1, by1) - torch.max(ay0, by0)).clamp(min=0)
    union = area_a + area_b - inter
    iou = (inter + 1e-6) / (union + 1e-6)
    hull = (torch.max(ax1, bx1) - torch.min(ax0, bx0)).clamp(min=0) * (torch.max(ay1, by1) - torch.min(ay0, by0)).clamp(min=0)
    return iou - ((hull - union) + 1e-6) / (hull + 1e-6)


HIP_MATCHER = True      # A/B switch: one-launch HIP cost matrix on the GPU
_BOX_ERRORS = {}      # device -> 0-dim int32 counter of malformed-box events (GPU path: recorded, never trapped)


def _assert_well_formed(xyxy, what):
    """The reference asserts x1 >= x0 and y1 >= y0 on the host (util/box_ops.py:75-76), which costs a device sync per
    call.  On the GPU the violation is COUNTED in a device-side flag instead (no sync, no device trap -- a trapping
    assert would take the process down with a GPU core dump); `raise_if_malformed_boxes()` turns it into the
    reference's AssertionError at the caller's next natural sync point.  On the CPU it asserts immediately."""
    ok = (xyxy[..., 2:] >= xyxy[..., :2]).all()
    if xyxy.is_cuda:
        flag = _BOX_ERRORS.get(xyxy.device)
        if flag is None:
            if torch.cuda.is_current_stream_capturing():
                return
            flag = _BOX_ERRORS[xyxy.device] = torch.zeros((), dtype=torch.int32, device=xyxy.device)
        flag += (~ok).to(torch.int32)
    else:
        assert ok, f"error boxes: {what}"


def raise_if_malformed_boxes():
    """Host-side check of the device flags (one sync).  Call it where the training loop already synchronises."""
    for dev, flag in _BOX_ERRORS.items():
        n = int(flag.item())
        if n:
            flag.zero_()
            raise AssertionError(f"error boxes: {n} malformed (x1 < x0 or y1 < y0, or NaN) box set(s) seen on {dev}")


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, cost_mask: float = 1,
                 cost_dice: float = 1, cost_boundary: float = 1, num_classes: int = 1):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.cost_mask, self.cost_dice, self.cost_boundary = cost_mask, cost_dice, cost_boundary
        self.num_classes = num_classes
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.mask_out_stride = 2

    @torch.no_grad()
    def cost_matrix_stacked(self, logits, boxes, masks, targets):
        """Matching cost of every query against the clip's single target for Lr decoder layers at once.
        logits [Lr,B,T,q,K], boxes [Lr,B,T,q,4], masks [Lr,B,T,q,h,w] -> [Lr,B,q]."""
        with torch.autocast(device_type=masks.device.type, enabled=False):
            logits, boxes, masks = logits.float(), boxes.float(), masks.float()
            lr, bs, nf, nq, h, w = masks.shape
            gt, _ = nested_tensor_from_tensor_list([t["masks"] for t in targets], size_divisibility=32, split=False).decompose()
            gt = gt.to(masks)
            s = self.mask_out_stride
            im_h, im_w = gt.shape[-2:]
            gt = gt[:, :, s // 2::s, s // 2::s]
            assert gt.size(2) * s == im_h and gt.size(3) * s == im_w

            valid = torch.stack([t["valid"] for t in targets]).to(logits.dtype)                  # [B, T]
            if masks.is_cuda and HIP_MATCHER:
                return self._cost_matrix_hip(logits, boxes, masks, gt, valid, targets)
            prob = logits.sigmoid()                                                              # [Lr, B, T, q, K]
            alpha, gamma = 0.25, 2.0
            neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
            pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
            if self.num_classes == 1:
                cls = (pos - neg)[..., 0]                                                         # [Lr, B, T, q]
            else:
                ids = torch.stack([t["labels"] for t in targets])                                # [B, T]
                cls = torch.gather(pos - neg, 4, ids[None, :, :, None, None].expand(lr, -1, -1, nq, 1))[..., 0]
            cost_class = (cls * valid[None, :, :, None]).sum(2) / valid.sum(1)[None, :, None]    # mean over valid frames

            tb = torch.stack([t["boxes"] for t in targets]).to(boxes.dtype)                      # [B, T, 4]
            cost_bbox = (boxes - tb[None, :, :, None, :]).abs().sum(-1).mean(2)                  # [Lr, B, q]
            pb, tbx = box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tb)
            _assert_well_formed(pb, "predictions")
            _assert_well_formed(tbx, "targets")
            cost_giou = -_pairwise_giou_1(pb, tbx[None, :, :, None, :]).mean(2)

            x = masks.transpose(2, 3).flatten(3)                                                 # [Lr, B, q, T*h*w]
            g = gt.flatten(1)[None, :, None, :]                                                  # [1, B, 1, T*h*w]
            p = x.sigmoid()
            ce = F.binary_cross_entropy_with_logits(x, g.expand_as(x), reduction="none")
            p_t = p * g + (1 - p) * (1 - g)
            focal = (alpha * g + (1 - alpha) * (1 - g)) * ce * ((1 - p_t) ** gamma)
            cost_mask = focal.mean(3)
            cost_dice = -((2 * (p * g).sum(3) + 1) / (p.sum(-1) + g.sum(-1) + 1))
            return (self.cost_class * cost_class + self.cost_bbox * cost_bbox + self.cost_giou * cost_giou
                    + self.cost_mask * cost_mask + self.cost_dice * cost_dice)

    def _cost_matrix_hip(self, logits, boxes, masks, gt, valid, targets):
        """One launch of csrc/matcher.hip for all layers, clips and queries."""
        from .._lib import check, lib
        lr, bs, nf, nq, h, w = masks.shape
        if masks.stride(-1) != 1 or masks.stride(-2) != w:
            masks = masks.contiguous()
        logits, boxes, gt = logits.contiguous(), boxes.contiguous(), gt.contiguous()
        tb = torch.stack([t["boxes"] for t in targets]).to(boxes.dtype).contiguous()
        labels = None if self.num_classes == 1 else torch.stack([t["labels"] for t in targets]).to(torch.int64).contiguous()
        flag = _BOX_ERRORS.get(masks.device)
        if flag is None and not torch.cuda.is_current_stream_capturing():
            flag = _BOX_ERRORS[masks.device] = torch.zeros((), dtype=torch.int32, device=masks.device)
        cost = torch.empty((lr, bs, nq), dtype=torch.float32, device=masks.device)
        sums = torch.empty((lr, bs, nq, 4), dtype=torch.float32, device=masks.device)
        sl, sb, st, sq = masks.stride()[:4]
        check(lib().ocpg_matcher_cost_f32(logits.data_ptr(), boxes.data_ptr(), masks.data_ptr(), sl, sb, st, sq, gt.data_ptr(), tb.data_ptr(),
                                          valid.contiguous().data_ptr(), None if labels is None else labels.data_ptr(), lr, bs, nf, nq,
                                          logits.shape[-1], h, w, float(self.cost_class), float(self.cost_bbox), float(self.cost_giou),
                                          float(self.cost_mask), float(self.cost_dice), sums.data_ptr(), cost.data_ptr(),
                                          None if flag is None else flag.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "ocpg_matcher_cost_f32")
        return cost

    @torch.no_grad()
    def cost_matrix(self, outputs, targets):
        """[B, q] matching cost of every query against the clip's single target."""
        return self.cost_matrix_stacked(outputs["pred_logits"][None], outputs["pred_boxes"][None], outputs["pred_masks"][None], targets)[0]

    @torch.no_grad()
    def match_stacked(self, logits, boxes, masks, targets):
        """-> int64 [Lr, B]: argmin query per clip and layer (one sync-free tensor program for all decoder layers)."""
        return self.cost_matrix_stacked(logits, boxes, masks, targets).argmin(dim=2)

    @staticmethod
    def as_indices(src):
        """[B] int64 -> the reference's list of (src_idx, tgt_idx) pairs (tgt is always 0: one object per clip)."""
        zero = torch.zeros(1, dtype=torch.int64, device=src.device)
        return [(src[i:i + 1], zero) for i in range(src.shape[0])]

    @torch.no_grad()
    def forward(self, outputs, targets):
        return self.as_indices(self.cost_matrix(outputs, targets).argmin(dim=1))


def build_matcher(args):
    if args.binary:
        num_classes = 1
    else:
        num_classes = {"ytvos": 65, "davis": 78, "a2d": 1, "jhmdb": 1}.get(args.dataset_file, 91)
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox, cost_giou=args.set_cost_giou,
                            cost_mask=args.set_cost_mask, cost_dice=args.set_cost_dice,
                            cost_boundary=args.set_cost_boundary, num_classes=num_classes)
