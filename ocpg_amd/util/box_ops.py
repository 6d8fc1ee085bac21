"""Box conversions and (generalised) IoU used by the matcher and the box losses (reference util/box_ops.py:29-85)."""
import torch


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


_TO_XYXY = {}


def box_cxcywh_to_xyxy(x):
    """[..., (cx, cy, w, h)] -> [..., (x0, y0, x1, y1)] (util/box_ops.py:9-13).

    fp32 GPU tensors: ONE matmul with a constant 4x4 (one kernel forward, one backward, instead of unbind + 4 elementwise
    + stack and their 9 backward nodes; the conversion is called ~10x per step on tiny tensors, the step is launch-bound).
    Bit-identical: every output is cx + (+-0.5) * w, the other two products are exact zeros."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 4:
        m = _TO_XYXY.get(x.device)
        if m is None:
            m = _TO_XYXY[x.device] = torch.tensor([[1, 0, 1, 0], [0, 1, 0, 1], [-0.5, 0, 0.5, 0], [0, -0.5, 0, 0.5]], dtype=torch.float32,
                                                  device=x.device)
        with torch.autocast(device_type="cuda", enabled=False):
            return torch.matmul(x, m)
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_iou(a, b):
    """Pairwise IoU [N,M] (with the reference's +1e-6 smoothing) and the union."""
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = box_area(a)[:, None] + box_area(b) - inter
    return (inter + 1e-6) / (union + 1e-6), union


def generalized_box_iou(a, b, check=True):
    """GIoU [N,M] for xyxy boxes. `check` keeps the reference's well-formedness assert (a host sync)."""
    if check:
        assert (a[:, 2:] >= a[:, :2]).all(), f"error boxes: {a} vs {b}."
        assert (b[:, 2:] >= b[:, :2]).all(), f"error boxes: {a} vs {b}."
    iou, union = box_iou(a, b)
    lt = torch.min(a[:, None, :2], b[:, :2])
    rb = torch.max(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - ((hull - union) + 1e-6) / (hull + 1e-6)
