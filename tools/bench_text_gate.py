"""Diagnostic (r4): the text gate (VisionLanguageFusionModule) forward + backward at level 0 of config #2, token-major vs batch-first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd.models.segmentation import VisionLanguageFusionModule
from ocpg_amd.models import amp_cache
dev = torch.device("cuda:0")
b, t, h, w, c, lk = 2, 5, 48, 80, 256, 12
fuse = VisionLanguageFusionModule(c, 8).to(dev)
vis = torch.randn(b, t * h * w, c, device=dev, requires_grad=True)
text = torch.randn(lk, b, c, device=dev, requires_grad=True)
pos = torch.randn(lk, b, c, device=dev)
pad = torch.zeros(b, lk, dtype=torch.bool, device=dev)
go = torch.randn(b, t * h * w, c, device=dev)


def run(bf):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if bf:
            out = fuse.forward_batch_first(vis, text, pad, pos)
        else:
            tok = vis.view(b, t, h, w, c).permute(1, 2, 3, 0, 4)
            out = fuse(visual=tok, text=text, text_key_padding_mask=pad, text_pos=pos)
            out = out.view(t, h, w, b, c).permute(3, 0, 1, 2, 4).reshape(b, t * h * w, c)
    out.backward(go)


for bf in (True, False, True, False):
    for _ in range(3):
        run(bf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(bf)
    e1.record()
    torch.cuda.synchronize()
    print("batch_first=%s: %.1f us per fwd+bwd" % (bf, e0.elapsed_time(e1) * 100))
