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
