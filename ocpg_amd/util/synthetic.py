"""Synthetic inputs of the benchmark / smoke / parity runs: no dataset is read anywhere."""
import torch


def synthetic_targets(b, t, h, w, device="cpu"):
    """The survey's synthetic target recipe: one centred rectangle per frame."""
    tg = []
    for _ in range(b):
        m = torch.zeros(t, h, w)
        y0, y1, x0, x1 = int(0.25 * h), int(0.5 * h), int(0.25 * w), int(0.5 * w)
        m[:, y0:y1, x0:x1] = 1.0
        tg.append({
            "size": torch.tensor([h, w]),
            "valid": torch.ones(t, dtype=torch.long),
            "labels": torch.zeros(t, dtype=torch.long),
            "boxes": torch.tensor([[0.375, 0.375, 0.25, 0.25]]).repeat(t, 1),
            "masks": m.clone(),
            "weights": 0.9 * m,
            "weak_masks": m.clone(),
        })
    return [{k: v.to(device) for k, v in d.items()} for d in tg]
