"""Folder readers of the Ref-YouTube-VOS / Ref-DAVIS layout and the training-clip dataset on top of them (reference
datasets/ytvos.py:41-243; layout: <root>/<split>/{JPEGImages,Annotations}/<video>/<frame>.{jpg,png}, <root>/<split>/meta.json,
<root>/meta_expressions/<split>/meta_expressions.json).

Frames are decoded with PIL straight into ONE uint8 tensor per clip ([T, 3, H, W]); everything after that is the tensor pipeline of
this package (build_target -> ClipPipeline).  Weak annotations (per-frame heat maps of every annotated object; h5 files in the
reference, datasets/ytvos.py:171-184) come through a `weak_loader(video, frame) -> (heatmaps [n, h, w], obj_ids)` callable: the default
reads `<root>/<split>/AnnotationsWeakly/<video>/<frame>.npz` (arrays `heatPoint`, `obj_ids`) and, when h5py is importable, the
reference's `.h5` files with the same two keys."""
import json
import os
import random
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from .clip_sampling import clips_of_video, sample_clip_indices
from .clip_transforms import ClipPipeline
from .targets import build_target, has_instance, weak_targets_from_heatmaps


def read_frames(folder: str, names: Sequence[str], ext: str = ".jpg") -> Tensor:
    """RGB frames -> uint8 [T, 3, H, W] (ytvos.py:168-169)."""
    from PIL import Image
    frames = [torch.from_numpy(np.asarray(Image.open(os.path.join(folder, n + ext)).convert("RGB")).copy()) for n in names]
    return torch.stack(frames).permute(0, 3, 1, 2).contiguous()


def read_object_masks(folder: str, names: Sequence[str], obj_id: int) -> Tensor:
    """Palette PNG label maps -> float [T, H, W] mask of object obj_id (ytvos.py:170,188-189)."""
    from PIL import Image
    maps = [torch.from_numpy(np.asarray(Image.open(os.path.join(folder, n + ".png")).convert("P")).copy()) for n in names]
    return (torch.stack(maps) == int(obj_id)).to(torch.float32)


def default_weak_loader(folder: str) -> Callable:
    def load(video: str, frame: str):
        base = os.path.join(folder, video, frame)
        if os.path.exists(base + ".npz"):
            z = np.load(base + ".npz")
            return torch.from_numpy(np.asarray(z["heatPoint"], dtype=np.float32)), [int(i) for i in z["obj_ids"]]
        if os.path.exists(base + ".h5"):
            import h5py                                    # not installed in every image: only needed for the reference's own files
            with h5py.File(base + ".h5", "r") as f:
                return torch.from_numpy(np.asarray(f["heatPoint"], dtype=np.float32)), [int(i) for i in f["obj_ids"]]
        raise FileNotFoundError(base + ".{npz,h5}")
    return load


class RefVideoIndex:
    """meta.json + meta_expressions.json of one split -> one record per (video, expression, anchor frame) -- ytvos.py:79-111.
    Category ids: the reference maps the 65 Ref-YouTube-VOS category names through a fixed table (datasets/categories.py); here they
    are numbered in sorted order of the names that occur in meta.json (the model is trained class-agnostic, `args.binary`,
    main.py:33-34, so only the distinction object / no object reaches the loss)."""

    def __init__(self, root: str, split: str, num_frames: int):
        self.root, self.split = root, split
        self.folder = os.path.join(root, split)
        with open(os.path.join(self.folder, "meta.json")) as f:
            objects = json.load(f)["videos"]
        with open(os.path.join(root, "meta_expressions", split, "meta_expressions.json")) as f:
            expressions = json.load(f)["videos"]
        names = sorted({o["category"] for v in objects.values() for o in v["objects"].values()})
        self.category_ids: Dict[str, int] = {n: i for i, n in enumerate(names)}
        self.videos: List[str] = list(expressions.keys())
        self.metas: List[dict] = []
        for vid in self.videos:
            frames = sorted(expressions[vid]["frames"])
            for exp_id, e in expressions[vid]["expressions"].items():
                for anchor in clips_of_video(len(frames), num_frames):
                    self.metas.append({"video": vid, "exp_id": exp_id, "exp": e["exp"], "obj_id": int(e["obj_id"]), "frames": frames,
                                       "frame_id": anchor, "category": objects[vid]["objects"][str(e["obj_id"])]["category"]})

    def __len__(self):
        return len(self.metas)


class RefVideoClips(torch.utils.data.Dataset):
    """Training samples `(clip [T, 3, h, w] float, target dict)` -- ytvos.py:123-243: sample the clip's frames around the anchor, read
    them and the referred object's masks, attach the weak-supervision maps, run the augmentation pipeline, and draw another record
    when the object is visible in no frame of the augmented clip."""

    def __init__(self, root: str, split: str, num_frames: int, pipeline: ClipPipeline, supervision: Optional[str] = "box",
                 weak_loader: Optional[Callable] = None, seed: Optional[int] = None, train: bool = True):
        self.index = RefVideoIndex(root, split, num_frames)
        self.num_frames, self.pipeline, self.supervision, self.train = num_frames, pipeline, supervision, train
        self.weak_loader = weak_loader or (default_weak_loader(os.path.join(self.index.folder, "AnnotationsWeakly")) if supervision else None)
        self.seed = seed

    def __len__(self):
        return len(self.index)

    def _rng(self, idx: int) -> random.Random:
        """Seeded: sample idx of a given seed is always the same augmented clip (workers do not share a global generator)."""
        return random.Random() if self.seed is None else random.Random(self.seed * 1000003 + idx)

    def __getitem__(self, idx: int):
        rng = self._rng(idx)
        while True:
            meta = self.index.metas[idx]
            frames = meta["frames"]
            picks = sample_clip_indices(len(frames), meta["frame_id"], self.num_frames, rng, train=self.train)
            names = [frames[i] for i in picks]
            clip = read_frames(os.path.join(self.index.folder, "JPEGImages", meta["video"]), names)
            masks = read_object_masks(os.path.join(self.index.folder, "Annotations", meta["video"]), names, meta["obj_id"])
            weights = weak_masks = weak_boxes = None
            if self.weak_loader is not None:
                w_list, m_list, b_list = [], [], []
                for n in names:
                    heat, ids = self.weak_loader(meta["video"], n)
                    if meta["obj_id"] in ids:                                   # ytvos.py:177-184
                        k = ids.index(meta["obj_id"])
                        m, b = weak_targets_from_heatmaps(heat, k)
                        w_list.append(heat[k]), m_list.append(m), b_list.append(b)
                    else:
                        z = torch.zeros(heat.shape[-2:])
                        w_list.append(z), m_list.append(z), b_list.append(torch.zeros(4))
                weights, weak_masks = torch.stack(w_list), torch.stack(m_list)
                weak_boxes = torch.stack(b_list) if self.supervision == "point" else None
            target = build_target(picks, self.index.category_ids[meta["category"]], masks, meta["exp"], weights, weak_masks, weak_boxes)
            clip, target = self.pipeline(clip, target, rng)
            if has_instance(target):
                return clip, target
            idx = rng.randint(0, len(self) - 1)                                   # ytvos.py:242-243


def expressions_of_split(root: str, split: str, exclude_split: Optional[str] = None) -> Dict[str, dict]:
    """videos -> {"frames": [...], "expressions": {exp_id: {"exp": ...}}} of a split, minus the videos that also appear in
    `exclude_split` (the competition's validation file lists the test videos too: inference_ytvos.py:79-89)."""
    with open(os.path.join(root, "meta_expressions", split, "meta_expressions.json")) as f:
        data = json.load(f)["videos"]
    if exclude_split is not None:
        with open(os.path.join(root, "meta_expressions", exclude_split, "meta_expressions.json")) as f:
            drop = set(json.load(f)["videos"].keys())
        data = {k: v for k, v in data.items() if k not in drop}
    return dict(sorted(data.items()))
