"""Host -> device staging of training batches beside the compute stream.

The model's input is small next to its compute: one step of BASELINE config #2 reads 2 clips x 5 frames x 3 x 384 x 640 = 7.4 MB as
uint8 (29.5 MB as normalised fp32) -- 0.13 ms (0.54 ms) of a PCIe 5.0 x16 link against a 40 ms step.  So clips cross the link as
DECODED uint8 frames from pinned memory on a separate HIP stream while the previous step computes, and are resized / normalised on
the GPU by the tensor pipeline of this package (clip_transforms works on any device).  The reference does all of that on the host
in PIL and uploads fp32 (datasets/ytvos.py:236-237, engine.py:41-44: `samples.to(device)` on the compute stream)."""
from typing import Callable, Iterable, Optional

import torch

from ..util.misc import targets_to


class DevicePrefetcher:
    """Iterates `loader` (batches `(clips, targets)`: clips = a tensor, a NestedTensor, or a list of tensors) one batch ahead and yields
    `(clips, captions, targets)` on `device`: batch i+1 is copied on a side stream while the caller works on batch i.  `on_device(clips, targets)`
    (optional) runs on the side stream right after the copy (e.g. GPU-side normalisation).  On a CPU device it is a plain pass-through
    that still applies `on_device`."""

    def __init__(self, loader: Iterable, device, on_device: Optional[Callable] = None):
        self.loader, self.device, self.on_device = loader, torch.device(device), on_device
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _upload(self, batch):
        clips, targets = batch

        def move(x):
            if torch.is_tensor(x):
                if self.stream is not None and not x.is_pinned() and x.device.type == "cpu":
                    x = x.pin_memory()
                return x.to(self.device, non_blocking=True)
            if isinstance(x, (list, tuple)):
                return type(x)(move(v) for v in x)
            if hasattr(x, "tensors") and hasattr(x, "mask"):                  # NestedTensor: keep the mask's valid-extent tag
                mask = None if x.mask is None else move(x.mask)
                if mask is not None and hasattr(x.mask, "_ocpg_key"):
                    mask._ocpg_key = x.mask._ocpg_key
                return type(x)(move(x.tensors), mask)
            return x.to(self.device) if hasattr(x, "to") else x
        captions = [t["caption"] for t in targets if "caption" in t]           # engine.py:42-44: captions are read before targets_to
        clips = move(clips)
        targets = targets_to(targets, self.device)
        if self.on_device is not None:
            clips, targets = self.on_device(clips, targets)
        return clips, captions, targets

    def _stage(self, it):
        try:
            batch = next(it)
        except StopIteration:
            return None
        if self.stream is None:
            return self._upload(batch), None
        with torch.cuda.stream(self.stream):
            out = self._upload(batch)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return out, ready

    def __iter__(self):
        it = iter(self.loader)
        staged = self._stage(it)
        while staged is not None:
            (clips, captions, targets), ready = staged
            if ready is not None:
                torch.cuda.current_stream(self.device).wait_event(ready)
                for t in _tensors(clips) + _tensors(targets):
                    t.record_stream(torch.cuda.current_stream(self.device))      # the side stream's allocator must not recycle them early
            staged = self._stage(it)                                              # next batch goes up while this one is consumed
            yield clips, captions, targets

    def __len__(self):
        return len(self.loader)


def _tensors(obj):
    if torch.is_tensor(obj):
        return [obj] if obj.is_cuda else []
    if isinstance(obj, dict):
        return [t for v in obj.values() for t in _tensors(v)]
    if isinstance(obj, (list, tuple)):
        return [t for v in obj for t in _tensors(v)]
    out = []
    for name in ("tensors", "mask"):
        v = getattr(obj, name, None)
        if torch.is_tensor(v) and v.is_cuda:
            out.append(v)
    return out
