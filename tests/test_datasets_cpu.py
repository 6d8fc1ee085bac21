"""Input side: clip sampling, the targets schema and the clip augmentations against the reference
(datasets/ytvos.py:22-283, datasets/transforms_video.py): the target arithmetic against tests/golden/clip_transforms.npz (the
reference's own functions run on seeded clips), the randomised parts (sampling, pipeline composition) as hand-derived cases of its
rules, the image filter against PIL."""
import os
import random

import pytest
import torch
import torch.nn.functional as F

from ocpg_amd.datasets import build_target, check_target, clips_of_video, eval_pipeline, mask_bounding_box, sample_clip_indices, train_pipeline
from ocpg_amd.datasets import clip_transforms as ct
from ocpg_amd.datasets.targets import has_instance
from ocpg_amd.util.misc import collate_fn


def _clip_and_target(t=5, h=360, w=640, empty_last=True):
    g = torch.Generator().manual_seed(3)
    clip = torch.randint(0, 256, (t, 3, h, w), dtype=torch.uint8, generator=g)
    masks = torch.zeros(t, h, w)
    masks[: t - 1 if empty_last else t, 100:200, 300:420] = 1
    return clip, build_target(range(t), 7, masks, "The  Cat on the LEFT,  upright", weights=torch.rand(t, 45, 80, generator=g),
                              weak_masks=torch.rand(t, 45, 80, generator=g))


def test_anchor_frames_and_clip_sampling_rules():
    assert clips_of_video(11, 5) == [0, 5, 10]                                   # ytvos.py:101
    assert sample_clip_indices(30, 12, 1, random.Random(0)) == [12]
    for seed in range(50):
        rng = random.Random(seed)
        vid_len, fid, n = rng.randint(1, 40), 0, rng.choice([2, 3, 5, 8])
        fid = rng.randrange(vid_len)
        idx = sample_clip_indices(vid_len, fid, n, random.Random(seed), train=True)
        assert len(idx) == max(n, 3) if n != 1 else 1                            # num_frames 2 still yields the 3 local frames (:136-140)
        assert fid in idx and all(0 <= i < vid_len for i in idx)
        assert idx == sorted(idx) or idx == sorted(idx, reverse=True)            # random reverse (:160-162)
        near = [i for i in idx if i != fid and abs(i - fid) <= 3]
        assert len(near) >= min(2, vid_len - 1) or vid_len <= 2 or fid in (0, vid_len - 1)
        assert idx == sample_clip_indices(vid_len, fid, n, random.Random(seed), train=True)      # reproducible
    # enough frames outside the local window: the extra frames all come from outside [min, max] of the three local ones (:144-149)
    idx = sample_clip_indices(100, 50, 8, random.Random(1), train=False)
    local = [i for i in idx if 47 <= i <= 53]
    assert len(local) >= 3 and all(i <= min(local) or i >= max(local) for i in idx if i not in local)
    # a video shorter than what is missing: every frame at least once (:154-157)
    idx = sample_clip_indices(3, 1, 8, random.Random(2), train=False)
    assert len(idx) == 8 and set(idx) == {0, 1, 2} and idx == sorted(idx)
    # evaluation never reverses
    assert all(sample_clip_indices(20, 5, 5, random.Random(s), train=False) == sorted(sample_clip_indices(20, 5, 5, random.Random(s), train=False))
               for s in range(20))


def test_target_schema_of_a_clip():
    clip, tg = _clip_and_target()
    assert set(tg) == {"frames_idx", "labels", "boxes", "masks", "valid", "caption", "orig_size", "size", "weights", "weak_masks"}
    assert tg["caption"] == "the cat on the left, upright"                        # lower-cased, single spaces (:126)
    assert tg["boxes"][0].tolist() == [300.0, 100.0, 419.0, 199.0]                 # inclusive last row / column (:113-119,191-193)
    assert tg["boxes"][4].tolist() == [0.0, 0.0, 0.0, 0.0] and tg["valid"].tolist() == [1, 1, 1, 1, 0]
    assert tg["labels"].tolist() == [7] * 5 and tg["orig_size"].tolist() == [360, 640] == tg["size"].tolist()
    assert tg["weights"].shape == (5, 360, 640)
    assert has_instance(tg)
    m = torch.zeros(6, 9)
    m[2, 3] = 1
    assert mask_bounding_box(m).tolist() == [3.0, 2.0, 3.0, 2.0] and mask_bounding_box(torch.zeros(4, 4)).tolist() == [0.0] * 4
    # weak maps go to the frame size with align_corners=True: the corner values survive (:231-233)
    w = torch.arange(12.0).view(1, 3, 4)
    tg2 = build_target([0], 1, torch.ones(1, 9, 16), "x", weights=w)
    assert tg2["weights"][0, 0, 0] == 0 and tg2["weights"][0, -1, -1] == 11 and tg2["weights"][0, 0, -1] == 3
    # point supervision: the weak box replaces the mask box on visible frames only (:194-198)
    wb = torch.tensor([[1.0, 2.0, 3.0, 4.0]] * 5)
    tg3 = build_target(range(5), 7, tg["masks"], "x", weak_boxes=wb)
    assert tg3["boxes"][0].tolist() == [1.0, 2.0, 3.0, 4.0] and tg3["boxes"][4].tolist() == [0.0] * 4


def test_resize_rule_and_resized_targets():
    # shorter side -> size, unless the longer side would pass max_size (transforms_video.py:214-240); truncation of the long side
    assert ct.resize_size(480, 854, 360, 640) == (360, 640)
    assert ct.resize_size(360, 640, 360, 640) == (360, 640)
    assert ct.resize_size(720, 1280, 512, 640) == (360, 640)                      # 512 * 1280 / 720 > 640 -> size = round(640 * 720 / 1280)
    assert ct.resize_size(500, 300, 400, None) == (666, 400)                      # int(400 * 500 / 300)
    assert ct.resize_size(300, 500, 400, None) == (400, 666)
    assert ct.resize_size(100, 200, (50, 30), None) == (30, 50)                   # an explicit (w, h) pair
    clip, tg = _clip_and_target()
    out, t2 = ct.resize_clip(clip, tg, 288, 640)
    assert out.shape == (5, 3, 288, 512) and out.dtype == torch.float32 and t2["size"].tolist() == [288, 512]
    assert torch.allclose(t2["boxes"][0], tg["boxes"][0] * 0.8)
    assert t2["masks"].dtype == torch.bool and torch.equal(t2["masks"], F.interpolate(tg["masks"][:, None], (288, 512), mode="nearest")[:, 0] > 0.5)
    assert t2["weights"].shape == (5, 288, 512) and tg["size"].tolist() == [360, 640]          # the input dict is not modified
    # the image filter: identity at the same size, constant images stay constant, a 2x shrink of a checkerboard averages it
    same, _ = ct.resize_clip(clip, None, 360, 640)
    assert same is clip
    board = (torch.arange(8)[:, None] + torch.arange(8)[None]) % 2 * 255.0
    small, _ = ct.resize_clip(board.expand(1, 3, 8, 8).to(torch.uint8), None, (4, 4))
    assert torch.allclose(small[..., 1:3, 1:3], torch.full((1, 3, 2, 2), 0.5), atol=1e-6) and (small - 0.5).abs().max() < 0.02
    # against PIL's bilinear resize (what torchvision's F.resize runs on the reference's PIL frames): equal to PIL's uint8 rounding
    import numpy as np
    from PIL import Image
    g = torch.Generator().manual_seed(5)
    for (h, w), (oh, ow) in (((97, 131), (60, 81)), ((64, 48), (100, 75)), ((360, 640), (288, 512))):
        # smooth + noisy content
        img = (torch.rand(3, h, w, generator=g) * 60 + torch.linspace(0, 190, w)[None, None, :]).to(torch.uint8)
        ours, _ = ct.resize_clip(img[None], None, (ow, oh))
        pil = np.asarray(Image.fromarray(img.permute(1, 2, 0).numpy()).resize((ow, oh), Image.BILINEAR)).astype(np.float32)
        d = np.abs(ours[0].permute(1, 2, 0).numpy() * 255.0 - pil)
        assert d.max() <= 1.01 and d.mean() <= 0.4, (d.max(), d.mean())      # PIL rounds to uint8 after each of its two passes


def test_crop_flip_check_and_normalise():
    clip, tg = _clip_and_target()
    out, t2 = ct.crop_clip(clip, tg, (150, 350, 120, 200))                         # top, left, height, width
    assert out.shape == (5, 3, 120, 200) and t2["size"].tolist() == [120, 200] and t2["masks"].shape == (5, 120, 200)
    assert t2["boxes"][0].tolist() == [0.0, 0.0, 69.0, 49.0] and t2["area"][0].item() == 69.0 * 49.0      # shifted and clipped (:135-141)
    assert torch.equal(t2["masks"], tg["masks"][:, 150:270, 350:550])
    # a window that misses the object: zero-area boxes -> invalid frames, boxes zeroed (:38-53)
    _, t3 = ct.crop_clip(clip, tg, (0, 0, 90, 200))
    t3 = check_target(t3)
    assert t3["valid"].tolist() == [0] * 5 and float(t3["boxes"].abs().sum()) == 0 and not has_instance(t3)
    assert check_target(dict(t2))["valid"].tolist() == [1, 1, 1, 1, 0]
    # horizontal flip: x' = w - x with the corners swapped (:168-174); an involution on everything
    f1, tf = ct.hflip_clip(clip, tg)
    assert tf["boxes"][0].tolist() == [640 - 419.0, 100.0, 640 - 300.0, 199.0]
    f2, tb = ct.hflip_clip(f1, tf)
    assert torch.equal(f2, clip) and torch.equal(tb["boxes"], tg["boxes"]) and torch.equal(tb["masks"], tg["masks"])
    assert torch.equal(tf["weights"], tg["weights"].flip(-1))
    assert ct.swap_left_right("the left cat, right of the upright lefty") == "the right cat, left of the upleft righty"      # :582-583
    # normalisation: (x / 255 - mean) / std; boxes -> cxcywh / (w, h, w, h) (:653-675)
    x, tn = ct.normalize_clip(clip, tg)
    px = clip[2, 1, 17, 33].item()
    assert abs(x[2, 1, 17, 33].item() - (px / 255.0 - 0.456) / 0.224) < 1e-6
    assert torch.allclose(tn["boxes"][0], torch.tensor([359.5 / 640, 149.5 / 360, 119.0 / 640, 99.0 / 360]))
    assert torch.equal(tn["masks"], tg["masks"])


@pytest.mark.parametrize("seed", range(6))
def test_training_pipeline_end_to_end(seed):
    """ytvos.py:256-275 on a clip: consistent targets whatever branch the seed takes; same seed -> same sample; collate pads to /32."""
    clip, tg = _clip_and_target()
    pipe = train_pipeline(max_size=640)
    x, t = pipe(clip, tg, random.Random(seed))
    x2, t2 = pipe(clip, tg, random.Random(seed))
    assert torch.equal(x, x2) and torch.equal(t["boxes"], t2["boxes"]) and t["caption"] == t2["caption"]
    h, w = x.shape[-2:]
    assert x.dtype == torch.float32 and x.shape[:2] == (5, 3) and t["size"].tolist() == [h, w] and t["masks"].shape == (5, h, w)
    assert min(h, w) in ct.TRAIN_SCALES or max(h, w) <= 640
    assert t["valid"].tolist()[4] == 0 and t["boxes"].min() >= 0 and t["boxes"].max() <= 1
    assert t["caption"] in ("the cat on the left, upright", "the cat on the right, upleft")
    flipped = t["caption"] != tg["caption"]
    # the normalised box of a visible frame still frames that frame's mask (to the half-pixel the inclusive corners cost)
    for k in range(4):
        if t["valid"][k]:
            ys, xs = torch.where(t["masks"][k])
            cx, cy, bw, bh = (t["boxes"][k] * torch.tensor([w, h, w, h])).tolist()
            assert abs((xs.min().item() + xs.max().item() + 1) / 2 - (cx + (0.5 if not flipped else 0.5))) <= 2.5
            assert abs((ys.min().item() + ys.max().item() + 1) / 2 - (cy + 0.5)) <= 2.5
            assert abs((xs.max() - xs.min()).item() - bw) <= 3 and abs((ys.max() - ys.min()).item() - bh) <= 3
    # two differently augmented clips -> one padded batch (util/misc.py:299-379)
    xb, tb = pipe(clip, tg, random.Random(seed + 100))
    samples, targets = collate_fn([(x, t), (xb, tb)])
    H, W = samples.tensors.shape[-2:]
    assert H % 32 == 0 and W % 32 == 0 and samples.tensors.shape[:3] == (2, 5, 3) and len(targets) == 2
    assert not samples.mask[0, :, :h, :w].any() and samples.mask[0, :, h:, :].all() and samples.mask[0, :, :, w:].all()
    assert torch.equal(samples.tensors[0, :, :, :h, :w], x)


def test_evaluation_pipeline():
    clip = torch.randint(0, 256, (3, 3, 480, 854), dtype=torch.uint8)
    x, t = eval_pipeline()(clip, None)
    assert x.shape == (3, 3, 360, 640) and t is None                              # ytvos.py:278-282
    x, _ = eval_pipeline()(torch.rand(2, 3, 360, 640), None)
    assert x.shape == (2, 3, 360, 640)


def test_target_arithmetic_against_the_reference_fixture():
    """tests/golden/clip_transforms.npz = the reference's own datasets/transforms_video.py (resize with its size rule, crop + Check,
    hflip + caption swap, Normalize) and datasets/ytvos.py:22-38 (weight2mask) run on seeded clips (make_fixtures.py:
    gen_clip_transforms): boxes / areas / sizes / validity / masks / weak maps equal; resized pixels equal PIL's to its uint8 rounding."""
    from conftest import Golden
    from ocpg_amd.datasets import weak_targets_from_heatmaps
    gold = Golden("clip_transforms")
    a, meta = gold, gold.meta
    t = lambda k: gold[k]
    clip = t("frames").permute(0, 3, 1, 2).contiguous()                            # [T, 3, H, W] uint8
    base = {"boxes": t("boxes"), "masks": t("masks"), "valid": torch.tensor([1, 1, 0]), "caption": meta["caption"],
            "size": torch.tensor(list(clip.shape[-2:])), "weights": t("weights"), "weak_masks": t("weak_masks")}

    def same_targets(tag, got, maps=False):
        assert torch.allclose(got["boxes"], t(tag + "_boxes"), atol=1e-5), tag
        assert got["size"].tolist() == t(tag + "_size").tolist(), tag
        assert torch.equal(got["masks"].to(torch.uint8), t(tag + "_masks")), tag
        if tag + "_area" in a:
            assert torch.allclose(got["area"], t(tag + "_area"), rtol=1e-6), tag
        if tag + "_valid" in a:
            assert got["valid"].tolist() == t(tag + "_valid").tolist(), tag
        if maps:
            for k in ("weights", "weak_masks"):
                assert torch.allclose(got[k], t(f"{tag}_{k}"), atol=1e-6), (tag, k)

    for h, w, size, max_size, oh, ow in t("size_rule").tolist():                    # transforms_video.py:214-240
        assert ct.resize_size(h, w, size, None if max_size < 0 else max_size) == (oh, ow)
    for tag, case in meta["cases"].items():
        if tag.startswith("rs"):
            size = tuple(case["size"]) if isinstance(case["size"], list) else case["size"]
            out, got = ct.resize_clip(clip, dict(base), size, case["max_size"])
            assert list(out.shape[-2:]) == case["out_hw"], tag
            same_targets(tag, got, maps=tag == "rs72")
            d = (ct.as_float_clip(out) * 255.0 - t(tag + "_img").float()).abs()
            assert d.max() <= 1.01 and d.mean() <= 0.4, (tag, d.max(), d.mean())
        else:
            out, got = ct.crop_clip(clip, dict(base), tuple(case["region"]))
            got = check_target(got)
            same_targets(tag, got, maps=tag == "crop_in")
            assert torch.equal(out, t(tag + "_img")), tag
    out, got = ct.hflip_clip(clip, dict(base))
    same_targets("flip", got, maps=True)
    assert torch.equal(out, t("flip_img")) and ct.swap_left_right(meta["caption"]) == meta["flipped_caption"]
    out, got = ct.normalize_clip(clip, dict(base))
    same_targets("norm", got)
    assert torch.allclose(out, t("norm_img"), atol=2e-6)
    heat = t("heat")
    for k in range(3):                                                               # ytvos.py:22-38
        m, b = weak_targets_from_heatmaps(heat, k)
        assert torch.equal(m, t(f"w2m_mask{k}")) and torch.allclose(b, t(f"w2m_box{k}"), atol=1e-5), k
    # nothing beats the background plane -> empty mask, zero box
    m, b = weak_targets_from_heatmaps(heat * 0.1, 0)
    assert float(m.sum()) == 0 and b.tolist() == [0.0, 0.0, 0.0, 0.0]


def _write_tiny_dataset(root, videos=("vidA", "vidB"), n_frames=7, h=48, w=64):
    """A Ref-YouTube-VOS style tree with two objects per video; object 2 of vidB never appears (exercises the re-draw)."""
    import json
    import os
    import numpy as np
    from PIL import Image
    g = np.random.default_rng(0)
    meta, exps = {"videos": {}}, {"videos": {}}
    for v in videos:
        names = ["%05d" % (5 * i) for i in range(n_frames)]
        for sub in ("JPEGImages", "Annotations", "AnnotationsWeakly"):
            os.makedirs(os.path.join(root, "train", sub, v), exist_ok=True)
        for i, n in enumerate(names):
            Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "train", "JPEGImages", v, n + ".jpg"))
            lab = np.zeros((h, w), dtype=np.uint8)
            lab[10:30, 5 + 2 * i:25 + 2 * i] = 1
            if v == "vidA":
                lab[35:45, 40:60] = 2
            ann = Image.fromarray(lab)
            ann.putpalette([0, 0, 0, 128, 0, 0, 0, 128, 0] + [0] * 759)
            ann.save(os.path.join(root, "train", "Annotations", v, n + ".png"))
            heat = np.zeros((2, h // 2, w // 2), dtype=np.float32)
            heat[0, 5:15, 3 + i:13 + i] = 0.9
            heat[1, 17:22, 20:30] = 0.8
            np.savez(os.path.join(root, "train", "AnnotationsWeakly", v, n + ".npz"), heatPoint=heat, obj_ids=np.array([1, 2]))
        meta["videos"][v] = {"objects": {"1": {"category": "zebra"}, "2": {"category": "ape"}}}
        exps["videos"][v] = {"frames": names, "expressions": {"0": {"exp": "The Zebra  on the left", "obj_id": "1"},
                                                              "1": {"exp": "an ape", "obj_id": "2"}}}
    os.makedirs(os.path.join(root, "meta_expressions", "train"), exist_ok=True)
    with open(os.path.join(root, "train", "meta.json"), "w") as f:
        json.dump(meta, f)
    with open(os.path.join(root, "meta_expressions", "train", "meta_expressions.json"), "w") as f:
        json.dump(exps, f)
    return names


def test_folder_dataset_yields_model_ready_clips(tmp_path):
    """datasets/ytvos.py:79-243 on a synthetic folder tree: index records, decoded clips, targets incl. the weak maps, the re-draw of
    samples whose object never shows, reproducibility per (seed, index), and the padded batch."""
    from ocpg_amd.datasets import ClipPipeline
    from ocpg_amd.datasets.video_folders import RefVideoClips, RefVideoIndex, read_frames, read_object_masks
    root = str(tmp_path)
    names = _write_tiny_dataset(root)
    index = RefVideoIndex(root, "train", num_frames=3)
    assert len(index) == 2 * 2 * 3 and index.category_ids == {"ape": 0, "zebra": 1}          # 2 videos x 2 expressions x anchors 0, 3, 6
    assert [m["frame_id"] for m in index.metas[:3]] == [0, 3, 6] and index.metas[0]["obj_id"] == 1
    clip = read_frames(os.path.join(root, "train", "JPEGImages", "vidA"), names[:2])
    assert clip.shape == (2, 3, 48, 64) and clip.dtype == torch.uint8
    m = read_object_masks(os.path.join(root, "train", "Annotations", "vidA"), names[:2], 2)
    assert m.shape == (2, 48, 64) and float(m[0, 35:45, 40:60].min()) == 1 and float(m.sum()) == 2 * 200
    small = ClipPipeline([lambda c, t, r: ct.resize_clip(c, t, 32, 64), lambda c, t, r: (c, check_target(t)), lambda c, t, r: ct.normalize_clip(c, t)])
    for supervision in ("box", "point"):
        ds = RefVideoClips(root, "train", 3, small, supervision=supervision, seed=7)
        x, t = ds[0]
        assert x.shape == (3, 3, 32, 42) and x.dtype == torch.float32 and t["caption"] == "the zebra on the left"
        assert t["masks"].shape == (3, 32, 42) and t["weights"].shape == (3, 32, 42) and t["weak_masks"].shape == (3, 32, 42)
        assert t["valid"].tolist() == [1, 1, 1] and t["labels"].tolist() == [1, 1, 1] and 0 in t["frames_idx"].tolist()
        assert float(t["boxes"].min()) >= 0 and float(t["boxes"].max()) <= 1
        x2, t2 = ds[0]
        assert torch.equal(x, x2) and torch.equal(t["boxes"], t2["boxes"])                 # same (seed, index) -> same sample
    # the ape is absent from vidB: those records re-draw until a clip shows its object (ytvos.py:240-243)
    ape_b = [i for i, m in enumerate(ds.index.metas) if m["video"] == "vidB" and m["obj_id"] == 2]
    x, t = ds[ape_b[0]]
    assert bool((t["valid"] == 1).any())
    samples, targets = collate_fn([ds[0], ds[4]])
    assert samples.tensors.shape == (2, 3, 3, 32, 64) and samples.mask.shape == (2, 3, 32, 64) and len(targets) == 2


def test_inference_writes_the_competition_png_layout(tmp_path):
    """inference_ytvos.py:168-245 / inference_davis.py:253-268 with a stand-in model: per expression one 0/255 PNG per frame at the
    ORIGINAL resolution, the 'validation minus test' video filter, and palette label maps of merged objects."""
    import json
    import numpy as np
    from PIL import Image
    from ocpg_amd import inference
    from ocpg_amd.datasets.video_folders import expressions_of_split
    root = str(tmp_path)
    names = _write_tiny_dataset(root)
    os.makedirs(os.path.join(root, "meta_expressions", "test"), exist_ok=True)
    with open(os.path.join(root, "meta_expressions", "test", "meta_expressions.json"), "w") as f:
        json.dump({"videos": {"vidB": {}}}, f)
    assert list(expressions_of_split(root, "train", exclude_split="test")) == ["vidA"]

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, clips, captions, targets):
            t, _, h, w = clips[0].shape
            assert (h, w) == (360, 480) and targets[0]["size"].tolist() == [360, 480]      # 48 x 64 frames: shorter side -> 360
            masks = torch.full((1, t, 2, h, w), -10.0)
            if "zebra" in captions[0].lower():
                masks[:, :, 1, : h // 2] = 10.0                                            # query 1: top half
            else:
                masks[:, :, 0, :, : w // 4] = 10.0                                         # query 0: left quarter
            logits = torch.tensor([[-3.0], [3.0]] if "zebra" in captions[0].lower() else [[3.0], [-3.0]]).expand(1, t, 2, 1)
            return {"pred_logits": logits, "pred_masks": masks}

    out = str(tmp_path / "out")
    n = inference.write_expression_masks(Stub(), root, "train", out, exclude_split="test")
    assert n == 2 * len(names)
    top = np.asarray(Image.open(os.path.join(out, "Annotations", "vidA", "0", names[3] + ".png")))
    left = np.asarray(Image.open(os.path.join(out, "Annotations", "vidA", "1", names[0] + ".png")))
    assert top.shape == (48, 64) and top.dtype == np.uint8 and set(np.unique(top)) == {0, 255}
    assert top[:23].min() == 255 and top[25:].max() == 0 and left[:, :15].min() == 255 and left[:, 17:].max() == 0
    assert not os.path.exists(os.path.join(out, "Annotations", "vidB"))
    labels = inference.merge_objects(torch.stack([torch.from_numpy(top / 255.0).float()[None], torch.from_numpy(left / 255.0).float()[None]]))
    inference.write_label_maps(labels, str(tmp_path / "davis" / "anno_0" / "vidA"))
    img = Image.open(str(tmp_path / "davis" / "anno_0" / "vidA" / "00000.png"))
    assert img.mode == "P" and set(np.unique(np.asarray(img))) == {0, 1, 2} and np.asarray(img)[40, 40] == 0
    assert torch.equal(torch.from_numpy(np.asarray(img).copy()), labels[0])
    assert inference.label_palette()[:12] == [0, 0, 0, 128, 0, 0, 0, 128, 0, 128, 128, 0]              # the VOC / DAVIS colours


def test_device_prefetcher_passes_batches_through_in_order():
    """datasets/prefetch.py off the GPU: every batch once, in order, captions split off before the targets lose their strings
    (engine.py:41-44), the NestedTensor's mask tag kept, `on_device` applied."""
    from ocpg_amd.datasets.prefetch import DevicePrefetcher
    clip, tg = _clip_and_target(t=2, h=40, w=56, empty_last=False)
    pipe = eval_pipeline(size=32, max_size=64)
    batches = []
    for k in range(3):
        x, t = ct.normalize_clip(*ct.resize_clip(clip, dict(tg, caption="clip %d" % k), 32 + 8 * k, 96))
        batches.append(collate_fn([(x, t), (x.flip(-1), dict(t))]))
    seen = []
    pf = DevicePrefetcher(batches, "cpu", on_device=lambda c, t: (c, [dict(d, seen=torch.tensor(1)) for d in t]))
    assert len(pf) == 3
    for k, (samples, captions, targets) in enumerate(pf):
        assert captions == ["clip %d" % k] * 2 and all("caption" not in t and int(t["seen"]) == 1 for t in targets)
        assert torch.equal(samples.tensors, batches[k][0].tensors) and hasattr(samples.mask, "_ocpg_key")
        seen.append(tuple(samples.tensors.shape[-2:]))
    assert seen == [(32, 64), (64, 64), (64, 96)] and pipe is not None


def test_precision_and_iou_metrics():
    """ocpg_amd/metrics.py against hand-computed cases of a2d_eval.py:29-67."""
    from ocpg_amd import metrics
    gt = torch.zeros(4, 10, 10, dtype=torch.bool)
    pred = torch.zeros(4, 10, 10, dtype=torch.bool)
    gt[0, :5] = True; pred[0, :5] = True                     # IoU 1
    gt[1, :, :6] = True; pred[1, :, 3:9] = True              # I = 30, U = 90 -> 1/3
    gt[2, :8] = True; pred[2, :6] = True                     # I = 60, U = 80 -> 0.75
    # instance 3: both empty -> (0 + eps) / (0 + eps) = 1
    iou, inter, union = metrics.mask_iou(pred, gt)
    assert torch.allclose(iou, torch.tensor([1.0, 1 / 3, 0.75, 1.0]), atol=1e-6) and inter.tolist() == [50, 30, 60, 0] and union.tolist() == [50, 90, 80, 0]
    m = metrics.precision_and_iou(pred, gt)
    assert m["P@0.5"] == 0.75 and m["P@0.7"] == 0.75 and m["P@0.8"] == 0.5 and m["P@0.9"] == 0.5
    assert abs(m["overall_iou"] - 140 / 220) < 1e-6 and abs(m["mean_iou"] - (1 + 1 / 3 + 0.75 + 1) / 4) < 1e-6
    # batches of different resolutions accumulate to the same numbers
    st = metrics.accumulate(None, pred[:2], gt[:2])
    st = metrics.accumulate(st, torch.nn.functional.pad(pred[2:].float(), (0, 4, 0, 2)).bool(), torch.nn.functional.pad(gt[2:].float(), (0, 4, 0, 2)).bool())
    assert metrics.summarize(st) == m
    # the prediction with the highest score is the one evaluated; ties keep the last
    scores = torch.tensor([[0.1, 0.9, 0.3], [0.5, 0.2, 0.5]])
    masks = torch.arange(2 * 3).view(2, 3, 1, 1).expand(2, 3, 2, 2)
    assert metrics.select_best_query(scores, masks)[:, 0, 0].tolist() == [1, 5]


def test_run_length_encoding_and_mask_postprocessing():
    """models/postprocessors.py: the COCO run-length format by round trip and hand-derived strings; the A2D post-process's un-pad ->
    resize -> inverted threshold (reference postprocessors.py:36-44) and the score-ordered RefCOCO masks (:124-141)."""
    import numpy as np
    from ocpg_amd.models import postprocessors as pp
    assert pp.rle_counts(np.zeros((2, 2))) == [4] and pp.rle_encode(np.zeros((2, 2)))["counts"] == b"4"
    assert pp.rle_counts(np.ones((2, 3))) == [0, 6] and pp.rle_encode(np.ones((2, 3)))["counts"] == b"06"
    m = np.array([[0, 1, 1], [0, 0, 1]], dtype=np.uint8)                          # column-major: 0 0 | 1 | 0 | 1 1
    assert pp.rle_counts(m) == [2, 1, 1, 2]
    assert pp.rle_encode(m)["counts"] == bytes([48 + 2, 48 + 1, 48 + 1, 48 + 1])   # the 4th count is stored as 2 - counts[1] = 1
    assert pp.rle_encode(np.zeros((40, 40)))["counts"] == bytes([48 + (1600 & 31) + 32, 48 + ((1600 >> 5) & 31) + 32, 48 + (1600 >> 10)])
    g = np.random.default_rng(0)
    for shape in ((1, 1), (7, 5), (64, 48), (33, 200)):
        for p in (0.02, 0.5, 0.97):
            x = (g.random(shape) < p).astype(np.uint8)
            r = pp.rle_encode(torch.from_numpy(x))
            assert r["size"] == list(shape) and np.array_equal(pp.rle_decode(r), x)
    blob = np.zeros((120, 90), dtype=np.uint8)
    blob[30:90, 20:70] = 1                                                         # long equal runs: negative deltas in the string
    assert np.array_equal(pp.rle_decode(pp.rle_encode(blob)), blob)
    # A2D: logits positive in the top-left quadrant of the un-padded region -> the INVERTED mask is False exactly there
    logits = torch.full((1, 1, 2, 16, 24), -8.0)
    logits[0, 0, 0, :6, :10] = 8.0
    out = {"pred_logits": torch.tensor([[[[2.0], [-1.0]]]]), "pred_masks": logits}
    res = pp.A2DSentencesPostProcess()(out, torch.tensor([[24, 40]]), torch.tensor([[12, 20]]))
    assert torch.allclose(res[0]["scores"], torch.tensor([2.0, -1.0]).sigmoid()) and res[0]["masks"].shape == (2, 1, 24, 40)
    assert not res[0]["masks"][0, 0, :11, :19].any() and res[0]["masks"][0, 0, 13:, :].all() and res[0]["masks"][1].all()
    assert np.array_equal(pp.rle_decode(res[0]["rle_masks"][0]), res[0]["masks"][0, 0].numpy().astype(np.uint8))
    # RefCOCO: masks come back ordered by descending score
    out2 = {"pred_logits": torch.tensor([[[[-1.0], [3.0]]]]), "pred_masks": logits, "pred_boxes": torch.rand(1, 1, 2, 4)}
    r2 = pp.PostProcessSegm()([{}], out2, torch.tensor([[16, 24]]), torch.tensor([[16, 24]]))
    assert r2[0]["masks"].shape == (2, 1, 16, 24) and int(r2[0]["masks"][0].sum()) == 0 and int(r2[0]["masks"][1].sum()) == 60
    import argparse
    assert isinstance(pp.build_postprocessors(argparse.Namespace(threshold=0.5, masks=True), "a2d"), pp.A2DSentencesPostProcess)
    assert set(pp.build_postprocessors(argparse.Namespace(threshold=0.5, masks=True), "ytvos")) == {"bbox", "segm"}


def test_evaluation_loop_with_a_stand_in_model():
    """engine.evaluate_referred_masks (reference engine.py:126-194 + a2d_eval.py:37-67): post-process -> best query -> metrics."""
    from ocpg_amd import engine
    from ocpg_amd.models.postprocessors import A2DSentencesPostProcess
    from ocpg_amd.util.misc import NestedTensor

    class Stub(torch.nn.Module):
        def forward(self, samples, captions, targets):
            b = samples.tensors.shape[0]
            masks = torch.full((b, 1, 2, 16, 24), 8.0)                 # the A2D post-process inverts: positive logits = background
            masks[:, 0, 1, :8, :12] = -8.0                              # query 1 segments the top-left quarter of the un-padded 16 x 24
            masks[:, 0, 0, :, :] = -8.0                                 # query 0 segments everything
            logits = torch.tensor([[-2.0], [2.0]]).expand(b, 1, 2, 1)   # query 1 scores highest
            return {"pred_logits": logits, "pred_masks": masks}

    def batch(gt_rows):
        gt = torch.zeros(32, 48, dtype=torch.bool)
        gt[:gt_rows, :24] = True
        t = {"caption": "x", "orig_size": torch.tensor([32, 48]), "size": torch.tensor([16, 24]), "gt_mask": gt}
        return NestedTensor(torch.zeros(1, 1, 3, 16, 24), torch.zeros(1, 1, 16, 24, dtype=torch.bool)), [t]
    # prediction = rows < 16, cols < 24 at 32 x 48; ground truths: identical (IoU 1) and half of it (IoU 0.5, not > 0.5)
    m = engine.evaluate_referred_masks(Stub(), [batch(16), batch(8)], A2DSentencesPostProcess(), "cpu")
    assert m["P@0.5"] == 0.5 and m["P@0.9"] == 0.5 and abs(m["mean_iou"] - 0.75) < 1e-6
    assert abs(m["overall_iou"] - (384 + 192) / (384 + 384)) < 1e-6
