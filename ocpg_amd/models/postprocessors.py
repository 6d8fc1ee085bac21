"""Evaluation-time output formatting.  Out of the hot path: COCO-RLE encoding for the
A2D / JHMDB / RefCOCO evaluators needs pycocotools and is not rebuilt; build_model still returns a third element so
the reference's drivers keep their call signature (models/postprocessors.py:145-152)."""
import torch
import torch.nn.functional as F
from torch import nn


class PostProcess(nn.Module):
    """Boxes/scores of the reference's `PostProcess` (postprocessors.py:57-93): per-image top query -> xyxy pixels."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        from ..util import box_ops
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        prob = logits.sigmoid()
        scores, labels = prob.flatten(1).max(-1)
        xyxy = box_ops.box_cxcywh_to_xyxy(boxes)
        h, w = target_sizes.unbind(1)
        scale = torch.stack([w, h, w, h], dim=1)
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, xyxy * scale[:, None, :])]


def build_postprocessors(args, dataset_name):
    print("\n **** BUILD POSTPROCESSOR FOR {}. **** \n".format(dataset_name)) if getattr(args, "verbose", False) else None
    if dataset_name in ("a2d", "jhmdb") or "coco" in dataset_name:
        return {"bbox": PostProcess()}
    return None
