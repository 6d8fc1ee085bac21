"""Frame sampling of a training clip (reference datasets/ytvos.py:99-111 for the anchors, :131-160 for the clip around an anchor)."""
import random
from typing import List, Optional


def clips_of_video(vid_len: int, num_frames: int) -> List[int]:
    """Anchor frames of one (video, expression): every num_frames-th frame (ytvos.py:101)."""
    return list(range(0, vid_len, num_frames))


def sample_clip_indices(vid_len: int, frame_id: int, num_frames: int, rng: Optional[random.Random] = None, train: bool = True,
                        reverse_p: float = 0.3) -> List[int]:
    """Indices (into the video's sorted frame list) of the num_frames frames of the clip anchored at frame_id:

      * the anchor, one frame 1..3 before and one 1..3 after it (clamped to the video) -- ytvos.py:133-140
      * the rest drawn without replacement from the frames OUTSIDE [min, max] of those three; when there are not enough of them, from
        the whole video; when the video itself is shorter than what is missing, every frame once plus random repeats -- :142-157
      * sorted; in training reversed with probability reverse_p (the reference draws that coin from numpy's global generator,
        :161; here the same `rng` serves both) -- :158-162

    `rng` is consumed in the reference's order (before, after, global sample, reverse coin), so a seeded `random.Random` reproduces
    its index lists as long as the coin is not looked at (reverse_p = 0)."""
    rng = rng or random
    idx = [frame_id]
    if num_frames != 1:
        before, after = rng.randint(1, 3), rng.randint(1, 3)
        idx += [max(0, frame_id - before), min(vid_len - 1, frame_id + after)]
        if num_frames > 3:
            everything = list(range(vid_len))
            outside = everything[:min(idx)] + everything[max(idx):]
            missing = num_frames - len(idx)
            if len(outside) > missing:
                idx += [outside[i] for i in rng.sample(range(len(outside)), missing)]
            elif vid_len >= missing:
                idx += [everything[i] for i in rng.sample(range(vid_len), missing)]
            else:
                extra = missing - vid_len
                # (the reference's sample() raises when even the repeats outnumber the frames -- videos that short do not occur in
                # its datasets; here they are drawn with replacement)
                repeats = rng.sample(range(vid_len), extra) if extra <= vid_len else [rng.randrange(vid_len) for _ in range(extra)]
                idx += [everything[i] for i in repeats + list(range(vid_len))]
    idx.sort()
    if train and reverse_p > 0 and rng.random() < reverse_p:
        idx.reverse()
    return idx
