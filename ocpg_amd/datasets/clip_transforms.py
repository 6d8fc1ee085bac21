"""Clip augmentations that keep the `targets` dict consistent (reference datasets/transforms_video.py; composition: datasets/ytvos.py:
245-283).  A clip is ONE tensor [T, 3, H, W] (uint8 0..255 or float 0..1), so a resize is one batched interpolation on the clip's
device instead of T PIL calls; the geometry of the targets (box arithmetic, nearest-neighbour masks, the size rule, the validity
re-check, the left/right caption swap) follows the reference line by line.  Pixel VALUES of a resized image are the same triangle
filter evaluated in float (PIL rounds every resized frame back to uint8): equal up to that rounding, not bit-equal.

Randomness comes from the `random.Random` handed to the pipeline; it is consumed in the reference's order (select, scale(s), crop
size w then h, crop position, flip)."""
import random
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from ..util.box_ops import box_xyxy_to_cxcywh
from .targets import check_target

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
TRAIN_SCALES = (288, 320, 352, 392, 416, 448, 480, 512)          # ytvos.py:253
_MAPS = ("weak_masks", "weights")


def as_float_clip(clip: Tensor) -> Tensor:
    """uint8 0..255 -> float32 0..1 (ToTensor, transforms_video.py:637-642); float clips pass through."""
    return clip.to(torch.float32) / 255.0 if clip.dtype == torch.uint8 else clip.to(torch.float32)


def resize_size(h: int, w: int, size, max_size: Optional[int] = None) -> Tuple[int, int]:
    """(h, w) after `resize(clip, size, max_size)`: the SHORTER side becomes `size` unless that pushes the longer side over
    max_size, in which case size shrinks to round(max_size * short / long); a (w, h) pair is taken as given -- transforms_video.py:
    214-240 (note the truncation, not rounding, of the longer side)."""
    if isinstance(size, (list, tuple)):
        return int(size[1]), int(size[0])
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def resize_clip(clip: Tensor, target: Optional[dict], size, max_size: Optional[int] = None):
    """transforms_video.py:211-299: frames bilinear (antialiased when shrinking, as PIL), boxes / area scaled by the per-axis ratio of
    the ACTUAL sizes, masks nearest-neighbour (and thresholded: bool afterwards), weak maps bilinear."""
    h, w = clip.shape[-2:]
    oh, ow = resize_size(h, w, size, max_size)
    out = clip if (oh, ow) == (h, w) else F.interpolate(as_float_clip(clip), (oh, ow), mode="bilinear", align_corners=False, antialias=True)
    if target is None:
        return out, None
    rw, rh = float(ow) / float(w), float(oh) / float(h)
    target = dict(target)
    if "boxes" in target:
        target["boxes"] = target["boxes"] * torch.as_tensor([rw, rh, rw, rh], dtype=torch.float32, device=target["boxes"].device)
    if "area" in target:
        target["area"] = target["area"] * (rw * rh)
    target["size"] = torch.tensor([oh, ow])
    if "masks" in target:
        m = target["masks"]
        target["masks"] = F.interpolate(m[:, None].float(), (oh, ow), mode="nearest")[:, 0] > 0.5 if m.shape[0] > 0 else torch.zeros((0, oh, ow))
    for key in _MAPS:
        if key in target:
            m = target[key]
            target[key] = (F.interpolate(m[:, None].float(), (oh, ow), mode="bilinear", align_corners=False)[:, 0] if m.shape[0] > 0
                           else torch.zeros((0, oh, ow)))
    return out, target


def crop_clip(clip: Tensor, target: dict, region: Tuple[int, int, int, int]):
    """region = (top, left, height, width); boxes are shifted, clipped to the window and their area recomputed -- transforms_video.py:
    97-158.  Validity is NOT touched here (check_target does that at the end of the branch)."""
    i, j, h, w = region
    out = clip[..., i:i + h, j:j + w]
    target = dict(target)
    target["size"] = torch.tensor([h, w])
    for key in ("masks",) + _MAPS:
        if key in target:
            target[key] = target[key][:, i:i + h, j:j + w]
    if "boxes" in target:
        b = target["boxes"]
        lim = torch.as_tensor([w, h], dtype=torch.float32, device=b.device)
        c = torch.min((b - torch.as_tensor([j, i, j, i], dtype=b.dtype, device=b.device)).reshape(-1, 2, 2), lim).clamp(min=0)
        target["boxes"] = c.reshape(-1, 4)
        target["area"] = (c[:, 1, :] - c[:, 0, :]).prod(dim=1)
    return out, target


def hflip_clip(clip: Tensor, target: dict):
    """transforms_video.py:161-189 (the caption swap lives in `swap_left_right`, as in RandomHorizontalFlip :580-584)."""
    w = clip.shape[-1]
    target = dict(target)
    if "boxes" in target:
        b = target["boxes"]
        target["boxes"] = b[:, [2, 1, 0, 3]] * torch.as_tensor([-1, 1, -1, 1], dtype=b.dtype, device=b.device) \
            + torch.as_tensor([w, 0, w, 0], dtype=b.dtype, device=b.device)
    for key in ("masks",) + _MAPS:
        if key in target:
            target[key] = target[key].flip(-1)
    return clip.flip(-1), target


def swap_left_right(caption: str) -> str:
    """'left' <-> 'right' (every occurrence, also inside words, as the reference's chained str.replace does) -- :582-583."""
    return caption.replace("left", "@").replace("right", "left").replace("@", "right")


def normalize_clip(clip: Tensor, target: Optional[dict], mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD):
    """ToTensor + Normalize (transforms_video.py:637-675): (x / 255 - mean) / std per channel; boxes xyxy in pixels -> cxcywh divided
    by the CURRENT (w, h, w, h)."""
    x = as_float_clip(clip)
    m = torch.as_tensor(mean, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    x = (x - m) / s
    if target is None:
        return x, None
    target = dict(target)
    h, w = x.shape[-2:]
    if "boxes" in target:
        b = target["boxes"]
        target["boxes"] = box_xyxy_to_cxcywh(b) / torch.tensor([w, h, w, h], dtype=torch.float32, device=b.device)
    return x, target


def random_size_crop(clip: Tensor, target: dict, min_size: int, max_size: int, rng):
    """transforms_video.py:328-337: width then height drawn in [min_size, min(side, max_size)], then the window's corner."""
    hh, ww = clip.shape[-2:]
    w = rng.randint(min_size, min(ww, max_size))
    h = rng.randint(min_size, min(hh, max_size))
    i = 0 if hh == h else rng.randint(0, hh - h)
    j = 0 if ww == w else rng.randint(0, ww - w)
    return crop_clip(clip, target, (i, j, h, w))


class ClipPipeline:
    """A fixed sequence of steps `(clip, target, rng) -> (clip, target)`; returns the normalised float clip [T, 3, h, w] and the
    targets the model / criterion consume."""

    def __init__(self, steps: List[Callable]):
        self.steps = list(steps)

    def __call__(self, clip: Tensor, target: Optional[dict], rng: Optional[random.Random] = None):
        rng = rng or random
        for step in self.steps:
            clip, target = step(clip, target, rng)
        return clip, target


def _resize_step(scales, max_size):
    return lambda clip, target, rng: resize_clip(clip, target, rng.choice(list(scales)), max_size)


def _check_step(clip, target, rng):
    return clip, check_target(target)


def _select_step(first: List[Callable], second: List[Callable], p: float = 0.5):
    def step(clip, target, rng):
        for s in (first if rng.random() < p else second):
            clip, target = s(clip, target, rng)
        return clip, target
    return step


def _flip_step(p: float = 0.5):
    def step(clip, target, rng):
        if rng.random() < p:
            target = dict(target)
            target["caption"] = swap_left_right(target["caption"])
            return hflip_clip(clip, target)
        return clip, target
    return step


def _normalize_step(clip, target, rng):
    return normalize_clip(clip, target)


def train_pipeline(max_size: int = 640, scales: Sequence[int] = TRAIN_SCALES) -> ClipPipeline:
    """ytvos.py:256-275: with probability 1/2 a plain multi-scale resize, else resize to 400 / 500 / 600 -> random 384..600 crop ->
    multi-scale resize; both end with the validity re-check; then the horizontal flip (with the caption swap) and the normalisation."""
    plain = [_resize_step(scales, max_size), _check_step]
    cropped = [_resize_step((400, 500, 600), None), lambda c, t, r: random_size_crop(c, t, 384, 600, r), _resize_step(scales, max_size), _check_step]
    return ClipPipeline([_select_step(plain, cropped), _flip_step(0.5), _normalize_step])


def eval_pipeline(size: int = 360, max_size: int = 640) -> ClipPipeline:
    """ytvos.py:278-282 (and the inference drivers' transform): shorter side 360, longer side at most 640, normalise."""
    return ClipPipeline([_resize_step((size,), max_size), _normalize_step])
