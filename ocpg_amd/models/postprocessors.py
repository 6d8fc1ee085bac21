"""Evaluation-time output formatting (reference models/postprocessors.py): scores / boxes / masks at the resolution of the raw frames and
the masks as COCO run-length encodings for the A2D-Sentences / JHMDB-Sentences / RefCOCO evaluators.

The reference encodes with pycocotools (`mask_util.encode`); that library is not in this image, so `rle_encode` restates the public
COCO format (column-major runs starting with a run of zeros, counts delta-coded against the count two places back from the fourth on,
5 payload bits per ASCII character offset by 48).  It is checked by round trips and hand-derived strings, NOT against pycocotools:
parity of the compressed string is unpinned."""
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


def rle_counts(mask: np.ndarray) -> List[int]:
    """[h, w] binary mask -> run lengths over the column-major pixel order, first run = zeros (possibly of length 0)."""
    flat = np.asarray(mask, dtype=np.uint8).reshape(-1, order="F")
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    return ([0] + counts) if flat[0] else counts


def rle_encode(mask) -> Dict:
    """-> {'size': [h, w], 'counts': bytes} (the dict pycocotools' `encode` returns for one mask)."""
    mask = mask.cpu().numpy() if torch.is_tensor(mask) else np.asarray(mask)
    counts = rle_counts(mask)
    out = bytearray()
    for i, c in enumerate(counts):
        x = c - counts[i - 2] if i > 2 else c
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return {"size": [int(mask.shape[0]), int(mask.shape[1])], "counts": bytes(out)}


def rle_decode(rle: Dict) -> np.ndarray:
    """Inverse of rle_encode -> uint8 [h, w]."""
    h, w = rle["size"]
    data, counts, p = rle["counts"], [], 0
    data = data.encode() if isinstance(data, str) else data
    while p < len(data):
        x, k, more = 0, 0, True
        while more:
            ch = data[p] - 48
            x |= (ch & 0x1F) << (5 * k)
            more = bool(ch & 0x20)
            p += 1
            k += 1
            if not more and (ch & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        flat[pos:pos + c] = val
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F")


class A2DSentencesPostProcess(nn.Module):
    """postprocessors.py:14-53: one annotated frame per sample (T = 1): per query the score, the mask un-padded to the augmented size,
    resized (bilinear) to the original size and binarised -- as `1 - (sigmoid > 0.5)`, the inversion the reference applies (:43) --
    plus its run-length encoding."""

    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    @torch.no_grad()
    def forward(self, outputs, orig_target_sizes, max_target_sizes):
        assert len(orig_target_sizes) == len(max_target_sizes)
        scores = outputs["pred_logits"][:, 0, :, 0].sigmoid()
        masks = outputs["pred_masks"][:, 0]
        predictions = []
        for s, m, resized, orig in zip(scores, masks, max_target_sizes, orig_target_sizes):
            mh, mw = int(resized[0]), int(resized[1])
            m = F.interpolate(m[:, :mh, :mw].unsqueeze(1).float(), size=tuple(int(v) for v in orig.tolist()), mode="bilinear", align_corners=False)
            m = ~(m.sigmoid() > 0.5)                                                  # [q, 1, H, W] bool
            predictions.append({"scores": s, "masks": m, "rle_masks": [rle_encode(q[0]) for q in m.cpu()]})
        return predictions


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


class PostProcessSegm(nn.Module):
    """postprocessors.py:96-143 (RefCOCO): masks re-ordered by descending query score, un-padded, resized to the original size,
    `sigmoid > 0.5`, with their run-length encodings; called after PostProcess on its `results`."""

    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    @torch.no_grad()
    def forward(self, results, outputs, orig_target_sizes, max_target_sizes):
        assert len(orig_target_sizes) == len(max_target_sizes)
        logits = outputs["pred_logits"].flatten(0, 1)                                  # [bt, q, k]
        masks = outputs["pred_masks"].flatten(0, 1)                                    # [bt, q, h, w]
        q = logits.shape[1]
        order = torch.topk(logits.sigmoid().flatten(1), k=q, dim=1, sorted=True)[1] // logits.shape[2]
        for i, (m, resized, orig) in enumerate(zip(masks, max_target_sizes, orig_target_sizes)):
            m = m[order[i]][:, :int(resized[0]), :int(resized[1])].unsqueeze(1).float()
            m = F.interpolate(m, size=tuple(int(v) for v in orig.tolist()), mode="bilinear", align_corners=False).sigmoid() > 0.5
            results[i]["masks"] = m.to(torch.uint8)
            results[i]["rle_masks"] = [rle_encode(x[0]) for x in m.cpu()]
        return results


def build_postprocessors(args, dataset_name):
    """postprocessors.py:145-152."""
    if dataset_name in ("a2d", "jhmdb"):
        return A2DSentencesPostProcess(threshold=getattr(args, "threshold", 0.5))
    post = {"bbox": PostProcess()}
    if getattr(args, "masks", False):
        post["segm"] = PostProcessSegm(threshold=getattr(args, "threshold", 0.5))
    return post
