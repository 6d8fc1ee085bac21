"""Host-side helpers of the hot path: NestedTensor, clip padding, inverse_sigmoid, tiny dist helpers.

Mirrors the names the reference's model code imports from util/misc.py (NestedTensor :380-402,
nested_tensor_from_tensor_list :318-352, nested_tensor_from_videos_list :354-377, inverse_sigmoid :560-564,
is_dist_avail_and_initialized / get_world_size) -- only what the per-clip path needs; logging, checkpoint
and all-gather helpers of that file are out of scope.
"""
from collections import OrderedDict
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import Tensor

# ---- memoisation of everything that depends ONLY on the padding mask -----------------------------------------------
# A batch's padding mask is "valid rectangle [0:h_i, 0:w_i] per image" and the collate functions below know (h_i, w_i) on
# the host.  They tag the mask tensor with that description (`_ocpg_key`); position encodings, per-level masks, valid
# ratios and the encoder's reference grid are pure functions of it and are computed ONCE per distinct description
# (reference: recomputed every forward, ~230 small kernels: position_encoding.py:52-84, backbone.py:97-103,
# deformable_transformer.py:125-131,268-281).  Untagged masks (a caller-built NestedTensor) take the uncached path.
MEMO_ENABLED = True
_MEMO = OrderedDict()
_MEMO_CAP = 512


def tag_rect_mask(mask: Tensor, valid_hw) -> Tensor:
    """Declare: mask[..., y, x] == (y >= h_i or x >= w_i) for image i in row-major order of the leading axes."""
    mask._ocpg_key = ("rect", int(mask.shape[-2]), int(mask.shape[-1]), tuple((int(h), int(w)) for h, w in valid_hw))
    return mask


def mask_key(mask):
    return getattr(mask, "_ocpg_key", None) if MEMO_ENABLED else None


def fully_valid(key) -> bool:
    """True when the mask a key describes has no padding at all (every image fills its map) -- known on the host."""
    if key is None:
        return False
    if key[0] == "rect":
        _, h, w, valid = key
        return all(vh == h and vw == w for vh, vw in valid)
    if key[0] == "resized":          # nearest resize of an all-False mask is all-False
        return fully_valid(key[1])
    return False


def memo(tag, key, device, compute):
    """compute() (tensors without autograd history, never modified in place downstream), cached on (tag, key, device)."""
    if key is None:
        return compute()
    k = (tag, key, str(device))
    v = _MEMO.get(k)
    if v is None:
        with torch.no_grad():
            v = compute()
        _MEMO[k] = v
        if len(_MEMO) > _MEMO_CAP:
            _MEMO.popitem(last=False)
    else:
        _MEMO.move_to_end(k)
    return v


def resize_mask(mask: Tensor, size) -> Tensor:
    """Nearest-neighbour resize of a [N,H,W] padding mask to a feature level (backbone.py:100-102), memoised."""
    size = (int(size[0]), int(size[1]))
    key = mask_key(mask)
    out = memo(("resize", size), key, mask.device, lambda: F.interpolate(mask[None].float(), size=size).to(torch.bool)[0])
    if key is not None:
        out._ocpg_key = ("resized", key, size)
    return out


class NestedTensor:
    """A padded batch plus its boolean padding mask (True = padding)."""

    def __init__(self, tensors: Tensor, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        mask = None if self.mask is None else self.mask.to(device)
        if mask is not None and hasattr(self.mask, "_ocpg_key"):
            mask._ocpg_key = self.mask._ocpg_key
        return NestedTensor(self.tensors.to(device), mask)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return f"NestedTensor({tuple(self.tensors.shape)})"


def _ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m if m > 1 else x


def nested_tensor_from_tensor_list(tensor_list: List[Tensor], size_divisibility: int = 1, split: bool = True) -> NestedTensor:
    """Pad a list of [C,H,W] maps (bottom/right, zeros) to a common size; H, W rounded up to `size_divisibility`.
    With split=True every element is first cut into 3-channel frames (a [T*3,H,W] clip -> T images)."""
    if split:
        tensor_list = [frame for t in tensor_list for frame in t.split(3, dim=0)]
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = max(t.shape[0] for t in tensor_list)
    h = _ceil_to(max(t.shape[1] for t in tensor_list), size_divisibility)
    w = _ceil_to(max(t.shape[2] for t in tensor_list), size_divisibility)
    ref = tensor_list[0]
    out = torch.zeros((len(tensor_list), c, h, w), dtype=ref.dtype, device=ref.device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=ref.device)
    for i, t in enumerate(tensor_list):
        out[i, : t.shape[0], : t.shape[1], : t.shape[2]] = t
        mask[i, : t.shape[1], : t.shape[2]] = False
    return NestedTensor(out, tag_rect_mask(mask, [(t.shape[1], t.shape[2]) for t in tensor_list]))


def nested_tensor_from_videos_list(videos_list: List[Tensor], size_divisibility: int = 1) -> NestedTensor:
    """Pad a list of [T,C,H,W] clips to [B,T,C,PH,PW] + mask [B,T,PH,PW]."""
    t = max(v.shape[0] for v in videos_list)
    c = max(v.shape[1] for v in videos_list)
    h = _ceil_to(max(v.shape[2] for v in videos_list), size_divisibility)
    w = _ceil_to(max(v.shape[3] for v in videos_list), size_divisibility)
    ref = videos_list[0]
    out = torch.zeros((len(videos_list), t, c, h, w), dtype=ref.dtype, device=ref.device)
    mask = torch.ones((len(videos_list), t, h, w), dtype=torch.bool, device=ref.device)
    for i, v in enumerate(videos_list):
        out[i, : v.shape[0], :, : v.shape[2], : v.shape[3]] = v
        mask[i, : v.shape[0], : v.shape[2], : v.shape[3]] = False
    valid_hw = [(v.shape[2], v.shape[3]) if j < v.shape[0] else (0, 0) for v in videos_list for j in range(t)]
    return NestedTensor(out, tag_rect_mask(mask, valid_hw))


def collate_fn(batch):
    """(clips, targets) pairs -> (NestedTensor padded to /32, tuple of targets); util/misc.py:299-307."""
    clips, targets = zip(*batch)
    return nested_tensor_from_videos_list(list(clips), size_divisibility=32), tuple(targets)


def inverse_sigmoid(x: Tensor, eps: float = 1e-5) -> Tensor:
    """log(x / (1 - x)) with both sides clamped at eps (util/misc.py:560-564).  One fused op (aten::logit clamps x to
    [eps, 1-eps] and evaluates the same quotient): bit-identical inside (eps, 1-eps).  At the clamped ends the reference
    evaluates log(eps / 1) resp. log(1 / eps) = -+11.5129 and this log(eps / (1 - eps)) resp. log((1 - eps) / fl(eps)) =
    -11.5129 / +11.5116 (fp32 rounding of 1 - eps): <= 1.4e-3 apart where the following sigmoid has slope 1e-5, i.e. <= 2e-8
    on any box coordinate.
    Gradient at the clamped ends DIFFERS from the reference: there x < eps gives log(eps / clamp(1 - x)) whose derivative
    through the un-clamped factor is 1 / (1 - x) ~ 1 (and 1 / x ~ 1 for x > 1 - eps), while logit's backward returns 0 outside
    [eps, 1 - eps].  It can only reach the decoder's level-0 reference points (the refined ones are detached,
    deformable_transformer.py:387) and only when a sigmoid output saturates below 1e-5 / above 1 - 1e-5."""
    return torch.logit(x, eps)


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def targets_to(targets, device):
    skip = {"nouns", "caption", "caption2", "caption3", "dataset_name", "original_id", "image_id"}
    return [{k: v.to(device) for k, v in t.items() if k not in skip} for t in targets]


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of scalar tensors in ONE collective (util/misc.py:162-186)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        vals = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(vals)
        if average:
            vals /= world
        return dict(zip(names, vals))
