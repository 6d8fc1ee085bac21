"""Video inference loop of the reference's DAVIS / Ref-YouTube-VOS drivers (reference
inference_davis.py:196-262, inference_ytvos.py has the same core): chop the video into clips of at most `clip_len` frames, run the
model in eval mode on each clip, keep the query with the highest mean score over the clip, un-pad, resize the mask logits to
the original resolution, sigmoid; `merge_objects` forms the multi-object label map.  Image loading / transforms / palette PNG
writing stay with the caller (datasets are out of scope)."""
import torch
import torch.nn.functional as F


@torch.no_grad()
def segment_video(model, frames, expression, clip_len=36, origin_size=None, amp_dtype=None):
    """frames [T,3,H,W] (already normalised, on the model's device); expression: str or PrecomputedText.
    Returns (logits [T,K] of the selected query, masks [T,H0,W0] in (0,1)); H0,W0 = origin_size or (H,W)."""
    model.eval()
    video_len, _, img_h, img_w = frames.shape
    out_h, out_w = origin_size if origin_size is not None else (img_h, img_w)
    size = torch.as_tensor([int(img_h), int(img_w)], device=frames.device)
    all_logits, all_masks = [], []
    for start in range(0, video_len, clip_len):                                  # inference_davis.py:203-206
        imgs = frames[start:start + clip_len]
        n = imgs.shape[0]
        with torch.autocast(device_type=frames.device.type, dtype=amp_dtype, enabled=amp_dtype is not None):
            outputs = model([imgs], [expression] if isinstance(expression, str) else expression, [{"size": size}])
        pred_logits = outputs["pred_logits"][0]                                  # [t, q, k]
        pred_masks = outputs["pred_masks"][0]                                    # [t, q, h, w]
        scores = pred_logits.sigmoid().mean(0).max(-1)[0]                        # [q]  (:228-231)
        best = scores.argmax(-1)
        masks = pred_masks[:, best].float()[None]                                # [1, t, h, w]
        masks = masks[:, :, :img_h, :img_w]                                      # unpad (:237)
        masks = F.interpolate(masks, size=(out_h, out_w), mode="bilinear", align_corners=False).sigmoid()[0]
        all_logits.append(pred_logits[:, best])
        all_masks.append(masks[:n])
    return torch.cat(all_logits, 0), torch.cat(all_masks, 0)


def merge_objects(masks, threshold=0.3, background=0.1):
    """masks [num_obj, T, h, w] in (0,1) -> uint8 [T, h, w]: 0 = background, i+1 = object i (inference_davis.py:253-259)."""
    masks = masks.clone()
    masks[masks < threshold] = 0.0
    bg = torch.full_like(masks[:1], background)
    return torch.cat([bg, masks], 0).argmax(0).to(torch.uint8)


def write_expression_masks(model, root, split, out_dir, videos=None, exclude_split=None, threshold=0.5, clip_len=None, device=None,
                           amp_dtype=None, transform=None):
    """Ref-YouTube-VOS style inference to files (reference inference_ytvos.py:168-245): for every video of the split and every
    expression, segment the whole video (one clip unless clip_len is given), threshold the probabilities at `threshold` and write
    <out_dir>/Annotations/<video>/<exp_id>/<frame>.png (8-bit, 0 / 255).  Returns the number of frames written."""
    import os

    from PIL import Image

    from .datasets.clip_transforms import eval_pipeline
    from .datasets.video_folders import expressions_of_split, read_frames
    data = expressions_of_split(root, split, exclude_split)
    transform = transform or eval_pipeline()
    device = device or next(model.parameters()).device
    written = 0
    for video in (videos if videos is not None else data.keys()):
        names = data[video]["frames"]
        raw = read_frames(os.path.join(root, split, "JPEGImages", video), names)
        origin = tuple(raw.shape[-2:])
        frames, _ = transform(raw.to(device), None)
        for exp_id, e in data[video]["expressions"].items():
            _, masks = segment_video(model, frames, e["exp"], clip_len=clip_len or len(names), origin_size=origin, amp_dtype=amp_dtype)
            binary = (masks > threshold).to(torch.uint8).mul(255).cpu().numpy()
            folder = os.path.join(out_dir, "Annotations", video, exp_id)
            os.makedirs(folder, exist_ok=True)
            for name, m in zip(names, binary):
                Image.fromarray(m).save(os.path.join(folder, name + ".png"))
                written += 1
    return written


def write_label_maps(labels, folder, palette=None):
    """uint8 [T, h, w] label maps (merge_objects) -> <folder>/00000.png ... as palette PNGs (inference_davis.py:262-268)."""
    import os

    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    palette = palette if palette is not None else label_palette()
    for f, lab in enumerate(labels.cpu().numpy()):
        img = Image.fromarray(lab)                      # 8-bit grey; putpalette turns it into a palette image with the same indices
        img.putpalette(palette)
        img.save(os.path.join(folder, "{:05d}.png".format(f)))


def label_palette(n=256):
    """The PASCAL-VOC / DAVIS label colour map (bit-interleaved RGB per index): what the reference copies out of a DAVIS annotation
    (inference_davis.py:156-157) when it has the dataset at hand."""
    pal = []
    for i in range(n):
        r = g = b = 0
        c = i
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal += [r, g, b]
    return pal
