"""Input side of the path: what turns an annotated video into the `(samples, captions, targets)` the model
and the criterion consume.

  clip_sampling.py    which frames of a video make one training clip (reference datasets/ytvos.py:131-160)
  targets.py          the per-clip `targets` dict, its invariants and the weak-supervision masks / boxes from heat maps
                      (datasets/ytvos.py:22-38,162-241, datasets/transforms_video.py:19-55)
  clip_transforms.py  resize / crop / flip / normalise of a clip together with its targets (datasets/transforms_video.py)
  video_folders.py    Ref-YouTube-VOS / Ref-DAVIS folder readers and the training-clip dataset (datasets/ytvos.py:41-243)
  prefetch.py         host -> device staging of batches on a side stream (engine.py:41-44 does it on the compute stream)

Everything works on tensors ([T, 3, H, W] clips, [T, H, W] masks) on whatever device they live on -- a clip can be decoded once, moved
to the GPU and augmented there -- and takes an explicit `random.Random` so that a worker's stream of augmentations is reproducible.
The folder readers decode with PIL into one uint8 tensor per clip; A2D-Sentences / JHMDB (video decoding through torchvision.io) are
not covered.
"""
from .clip_sampling import clips_of_video, sample_clip_indices
from .clip_transforms import ClipPipeline, eval_pipeline, train_pipeline
from .targets import build_target, check_target, mask_bounding_box, weak_targets_from_heatmaps

__all__ = ["clips_of_video", "sample_clip_indices", "ClipPipeline", "eval_pipeline", "train_pipeline", "build_target", "check_target",
           "mask_bounding_box", "weak_targets_from_heatmaps"]
