"""Vision-language fusion block and the mask loss library of the weakly-supervised criterion.

Reference: models/segmentation.py -- VisionLanguageFusionModule :95-113; sigmoid_focal_loss :134-160,
masked_ce_loss :173-200, dice_coefficient :203-211, generate_box_region_mask :224-238, proj_loss :253-277,
length_regularization / region_levelset / levelset_loss :279-315.  The unused pairwise / colour-similarity terms
(:375-513) are not on the training path (criterion.py never calls them) and are out of scope.
All losses are sync-free tensor programs (no Python loops over boxes, no .item()).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .attention import MultiheadAttention


class VisionLanguageFusionModule(nn.Module):
    """visual tokens attend to the text tokens; the attended text then GATES (multiplies) the visual input."""

    def __init__(self, d_model, nhead, dropout=0.0):
        super().__init__()
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)

    def forward(self, visual, text, text_key_padding_mask=None, text_pos=None, visual_pos=None):
        t, h, w, b, c = visual.shape
        visual = visual.reshape(t * h * w, b, c)
        q = visual if visual_pos is None else visual + visual_pos
        k = text if text_pos is None else text + text_pos
        return visual * self.multihead_attn(q, k, text, key_padding_mask=text_key_padding_mask)

    def forward_batch_first(self, visual, text, text_key_padding_mask=None, text_pos=None):
        """visual [b, L, c] (the channels-last map's own memory) -> [b, L, c], or None when the attention kernel does not serve the call."""
        k = text if text_pos is None else text + text_pos
        a = self.multihead_attn.forward_batch_first(visual, k, text, key_padding_mask=text_key_padding_mask)
        return None if a is None else visual * a


def dice_loss(inputs, targets, num_boxes):
    p = inputs.sigmoid().flatten(1)
    num = 2 * (p * targets).sum(1)
    den = p.sum(-1) + targets.sum(-1)
    return (1 - (num + 1) / (den + 1)).sum() / num_boxes


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    p = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def masked_ce_loss(inputs, weights, masks, box_regions=None, num_boxes=5, alpha=0.7, beta=0.3, thres=0.5):
    """Heat-map weighted BCE of the reference (incl. its double squashing: BCE-with-logits applied to sigmoid(x)*w).
    The heat-map is clamped to [beta, alpha], folded around `thres`, min-max normalised over the whole batch and
    forced to 1 outside the box region."""
    weight = (weights.clamp(min=beta, max=alpha) - thres).abs()
    lo, hi = weight.min(), weight.max()
    weight = (weight - lo) / (hi - lo + 1e-5)
    if box_regions is not None:
        weight = torch.where(box_regions == 0, torch.ones_like(weight), weight)
    loss = F.binary_cross_entropy_with_logits(inputs.sigmoid() * weight, masks * weight)
    return loss, weight


def dice_coefficient(x, target):
    n = x.size(0)
    x, target = x.reshape(n, -1), target.reshape(n, -1)
    inter = (x * target).sum(dim=1)
    union = (x ** 2.0).sum(dim=1) + (target ** 2.0).sum(dim=1) + 1e-5
    return 1.0 - (2 * inter / union)


def _py_slice_bounds(start, stop, size):
    """Python slice semantics start:stop on an axis of length `size`, vectorised (negative indices wrap)."""
    start = torch.where(start < 0, (start + size).clamp(min=0), start.clamp(max=size))
    stop = torch.where(stop < 0, (stop + size).clamp(min=0), stop.clamp(max=size))
    return start, stop


def generate_box_region_mask(boxes, mask_size, sizes):
    """boxes [m,4] xyxy normalised, sizes [m,2] (h,w) -> [m,H,W] float mask, 1 inside the integer-truncated box."""
    H, W = mask_size
    wh = torch.stack([sizes[:, 1], sizes[:, 0], sizes[:, 1], sizes[:, 0]], -1).to(boxes.device)
    sc = (boxes * wh).int()
    ok = ((sc[:, 3] - sc[:, 1]) > 0) & ((sc[:, 2] - sc[:, 0]) > 0)
    y0, y1 = _py_slice_bounds(sc[:, 1], sc[:, 3], H)
    x0, x1 = _py_slice_bounds(sc[:, 0], sc[:, 2], W)
    ys = torch.arange(H, device=boxes.device).view(1, H, 1)
    xs = torch.arange(W, device=boxes.device).view(1, 1, W)
    inside = (ys >= y0.view(-1, 1, 1)) & (ys < y1.view(-1, 1, 1)) & (xs >= x0.view(-1, 1, 1)) & (xs < x1.view(-1, 1, 1))
    return (inside & ok.view(-1, 1, 1)).float()


def proj_loss(inputs, box_regions, masks, num_boxes, with_mean_term=False):
    p = inputs.sigmoid()
    ly = dice_coefficient(p.max(dim=2, keepdim=True)[0], box_regions.max(dim=2, keepdim=True)[0])
    lx = dice_coefficient(p.max(dim=3, keepdim=True)[0], box_regions.max(dim=3, keepdim=True)[0])
    loss_max = (ly + lx).mean()
    if not with_mean_term:
        return loss_max
    m = masks.float()
    ly = dice_coefficient(p.mean(dim=2, keepdim=True), m.mean(dim=2, keepdim=True))
    lx = dice_coefficient(p.mean(dim=3, keepdim=True), m.mean(dim=3, keepdim=True))
    return loss_max + 0.1 * (ly + lx).mean()


def length_regularization(score):
    gh = (score[:, :, 1:, :] - score[:, :, :-1, :]).abs()
    gw = (score[:, :, :, 1:] - score[:, :, :, :-1]).abs()
    return gh.sum(dim=(1, 2, 3)) + gw.sum(dim=(1, 2, 3))


def region_levelset(score, target):
    """score [N,2,H,W] (fg, bg), target [N,C,H,W] -> [N] Chan-Vese style region energy."""
    fg, bg = score[:, 0:1], score[:, 1:2]
    c_in = (fg * target).sum((2, 3)) / fg.sum((2, 3)).clamp(min=0.00001)
    c_out = (bg * target).sum((2, 3)) / bg.sum((2, 3)).clamp(min=0.00001)
    e_in = (target - c_in[..., None, None]) ** 2
    e_out = (target - c_out[..., None, None]) ** 2
    return (e_in * fg + e_out * bg).sum((1, 2, 3)) / target.shape[1]


def levelset_loss(mask_logits, targets, box_mask_target):
    p = mask_logits.sigmoid()
    scores = torch.cat((p, 1.0 - p), dim=1) * box_mask_target
    pixels = box_mask_target.sum((1, 2, 3)).clamp(min=1)
    region = region_levelset(scores, targets * box_mask_target) / pixels
    length = 0.00001 * length_regularization(scores) / pixels
    return (region + length).mean()
