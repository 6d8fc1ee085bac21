"""Referring-segmentation metrics of the A2D-Sentences / JHMDB-Sentences evaluation (reference datasets/a2d_eval.py:29-67, fed by
engine.py:158-189): mask IoU, precision@K, overall IoU, mean IoU -- as one batched tensor program on whatever device the masks are on
(the reference decodes COCO RLE annotations one instance at a time on the host; the COCO-style AP numbers it also prints need
pycocotools and are not reproduced here)."""
from typing import Dict, Sequence

import torch
from torch import Tensor

THRESHOLDS = (0.5, 0.6, 0.7, 0.8, 0.9)


def mask_iou(pred: Tensor, gt: Tensor, eps: float = 1e-6):
    """pred, gt [N, H, W] (bool / 0-1) -> (iou [N], intersection [N], union [N]); (I + eps) / (U + eps): two empty masks count as
    IoU 1 -- a2d_eval.py:29-34."""
    p, g = pred.bool(), gt.bool()
    inter = (p & g).flatten(1).sum(1).to(torch.float32)
    union = (p | g).flatten(1).sum(1).to(torch.float32)
    return (inter + eps) / (union + eps), inter, union


def select_best_query(scores: Tensor, masks: Tensor) -> Tensor:
    """scores [N, Q], masks [N, Q, H, W] -> the mask of the highest-scoring query per instance [N, H, W] (a2d_eval.py:47-48: the
    prediction with the highest score; on equal scores the reference's stable sort keeps the LAST one, as here)."""
    q = scores.shape[1]
    best = q - 1 - scores.flip(1).argmax(dim=1)
    return masks[torch.arange(masks.shape[0], device=masks.device), best]


def precision_and_iou(pred: Tensor, gt: Tensor, thresholds: Sequence[float] = THRESHOLDS) -> Dict[str, float]:
    """One prediction and one ground-truth mask per instance ([N, H, W] each) -> {'P@0.5'..'P@0.9', 'overall_iou', 'mean_iou'}:
    precision@K = share of instances with IoU strictly above K, overall IoU = total intersection / total union, mean IoU = mean of
    the per-instance IoUs -- a2d_eval.py:37-67.  Instances of different sizes: call `accumulate` per batch and `summarize` once."""
    return summarize(accumulate(None, pred, gt), thresholds)


def accumulate(state, pred: Tensor, gt: Tensor):
    """Adds a batch to the running sums (state = None to start): (ious list, total intersection, total union)."""
    iou, inter, union = mask_iou(pred, gt)
    if state is None:
        return [iou], inter.sum(), union.sum()
    return state[0] + [iou], state[1] + inter.sum(), state[2] + union.sum()


def summarize(state, thresholds: Sequence[float] = THRESHOLDS) -> Dict[str, float]:
    ious = torch.cat(state[0])
    out = {"P@%s" % k: float((ious > k).float().mean()) for k in thresholds}
    out["overall_iou"] = float(state[1] / state[2]) if float(state[2]) > 0 else 0.0      # all masks empty: no union
    out["mean_iou"] = float(ious.mean())
    return out
