"""Diagnostic: forward-only (eval, DAVIS-style tail) throughput through ocpg_amd.inference.segment_video, bf16 autocast."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from ocpg_amd import inference
from ocpg_amd.models import build_model
from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
args = bench.model_args(dev, "resnet101", amp=True)
args.dataset_file = "davis"
model, _, _ = build_model(args)
model.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
T = int(os.environ.get("FRAMES", "72"))
frames = torch.randn(T, 3, 384, 640, device=dev)
text = PrecomputedText(torch.randn(1, 9, 768, device=dev), torch.randn(1, 768, device=dev), torch.zeros(1, 9, dtype=torch.bool, device=dev))
for clip in (36, 12):
    for _ in range(2): inference.segment_video(model, frames, text, clip_len=clip, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): inference.segment_video(model, frames, text, clip_len=clip, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("clip_len %2d: %d frames in %.1f ms -> %.0f frames/s" % (clip, T, dt * 1e3, T / dt))
