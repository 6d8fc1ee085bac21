"""The per-clip `targets` dict (reference datasets/ytvos.py:186-241) and the validity re-check the augmentations end with
(datasets/transforms_video.py:19-55).

Schema (T = frames of the clip; everything a tensor except the caption):
  frames_idx [T] int64        indices of the frames in the video
  labels     [T] int64        category id of the referred object
  boxes      [T, 4] float32   xyxy in pixels of the current size (cxcywh / size after `normalize`); zeros where the object is absent
  masks      [T, H, W]        float {0, 1} (bool after a resize, as in the reference)
  valid      [T]              1 where the object is visible in the frame
  caption    str              lower-cased, single spaces
  orig_size, size [2] int64   (h, w) of the raw frames / of the current (augmented) frames
  weights, weak_masks [T, H, W] float32   optional: the weak-supervision heat map and the mask derived from it (ytvos.py:171-184)
The matcher and the criterion read labels / boxes / masks / valid / weak_masks / weights (models/matcher.py:74-160, models/criterion.py:
46-226); the model reads size / caption / valid_indices."""
from typing import Optional

import torch
from torch import Tensor


def mask_bounding_box(mask: Tensor) -> Tensor:
    """xyxy of the pixels > 0 of one [H, W] mask, as INCLUSIVE pixel indices (x2, y2 = last row / column that holds a pixel), which is
    what the reference stores (ytvos.py:113-119,191-193); zeros for an empty mask (:198)."""
    rows = torch.any(mask > 0, dim=1)
    cols = torch.any(mask > 0, dim=0)
    if not bool(rows.any()):
        return torch.zeros(4, dtype=torch.float32, device=mask.device)
    ys, xs = torch.where(rows)[0], torch.where(cols)[0]
    return torch.stack([xs[0], ys[0], xs[-1], ys[-1]]).to(torch.float32)


def weak_targets_from_heatmaps(heatmaps: Tensor, instance_index: int, thres: float = 0.5):
    """Weak supervision of one frame (reference datasets/ytvos.py:22-38): heatmaps [n, h, w], one per annotated object -> the mask of
    the pixels where object `instance_index` beats every other object AND the constant background plane `thres` (arg-max over n + 1
    planes, first maximum wins), and the xyxy box around it built from its projections: extent = number of occupied columns / rows,
    centre = their centre of mass (so a mask with gaps gets a box tighter than its hull).  An empty mask gives the zero box."""
    n, h, w = heatmaps.shape
    planes = torch.cat([heatmaps, torch.full((1, h, w), thres, dtype=heatmaps.dtype, device=heatmaps.device)], dim=0)
    mask = (planes.argmax(dim=0) == instance_index).to(torch.float32)
    cols, rows = mask.amax(0), mask.amax(1)                        # occupied columns [w] / rows [h]
    bw, bh = cols.sum(), rows.sum()
    xs = torch.arange(w, dtype=torch.float32, device=mask.device)
    ys = torch.arange(h, dtype=torch.float32, device=mask.device)
    cx = (cols * xs).sum() / bw.clamp(min=1e-6)
    cy = (rows * ys).sum() / bh.clamp(min=1e-6)
    return mask, torch.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh])


def build_target(frames_idx, category_id: int, masks: Tensor, caption: str, weights: Optional[Tensor] = None,
                 weak_masks: Optional[Tensor] = None, weak_boxes: Optional[Tensor] = None) -> dict:
    """masks [T, H, W] (the referred object's binary masks on the raw frames) -> the targets dict of ytvos.py:215-229.

    Boxes come from the masks (clamped to the frame, :211-213) unless `weak_boxes` [T, 4] replaces them on the frames where the
    object is visible (point supervision, :194-195).  weights / weak_masks [T, h, w] at any resolution are brought to the frame size
    with align_corners=True bilinear interpolation (:231-233)."""
    masks = (masks > 0).to(torch.float32)
    t, h, w = masks.shape
    boxes = torch.stack([mask_bounding_box(m) for m in masks])
    valid = (masks.flatten(1).sum(1) > 0).to(torch.int64)
    if weak_boxes is not None:
        boxes = torch.where(valid[:, None].bool(), weak_boxes.to(boxes), boxes)
    boxes[:, 0::2] = boxes[:, 0::2].clamp(min=0, max=w)
    boxes[:, 1::2] = boxes[:, 1::2].clamp(min=0, max=h)
    target = {
        "frames_idx": torch.as_tensor(list(frames_idx), dtype=torch.int64),
        "labels": torch.full((t,), int(category_id), dtype=torch.int64),
        "boxes": boxes,
        "masks": masks,
        "valid": valid,
        "caption": " ".join(caption.lower().split()),
        "orig_size": torch.as_tensor([h, w]),
        "size": torch.as_tensor([h, w]),
    }
    for key, maps in (("weights", weights), ("weak_masks", weak_masks)):
        if maps is not None:
            maps = maps.to(torch.float32)
            if tuple(maps.shape[-2:]) != (h, w):
                maps = torch.nn.functional.interpolate(maps[None], (h, w), mode="bilinear", align_corners=True)[0]
            target[key] = maps
    return target


def check_target(target: dict) -> dict:
    """After cropping / resizing: a frame is valid while its box still has positive width and height (or, without boxes, while its
    mask still has a pixel); boxes of frames that lost the object become zeros -- transforms_video.py:38-53."""
    if "boxes" in target:
        corners = target["boxes"].reshape(-1, 2, 2)
        keep = torch.all(corners[:, 1, :] > corners[:, 0, :], dim=1)
        target["boxes"] = torch.where(keep[:, None], target["boxes"], torch.zeros_like(target["boxes"]))
    else:
        keep = target["masks"].flatten(1).any(1)
    target["valid"] = keep.to(torch.int32)
    return target


def has_instance(target: dict) -> bool:
    """The dataset re-draws a sample whose clip shows the object in no frame (ytvos.py:240-243)."""
    return bool(torch.any(target["valid"] == 1))
