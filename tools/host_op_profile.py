"""Diagnostic: HOST (CPU) self time per op over one bench step (torch.profiler), forward thread and autograd thread."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from ocpg_amd.models import build_model
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
args = bench.model_args(dev, "resnet101", amp=True)
model, crit, _ = build_model(args)
model.to(dev); crit.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(2, dev, 42)
step = bench.EagerStep(model, model, crit, opt, make_samples, text, targets, args, torch.bfloat16)
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print("unprofiled: %.2f ms/step" % ((time.perf_counter() - t0) / 5 * 1e3))
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(e.self_cpu_time_total for e in ka)
print("total self CPU %.1f ms over %d op calls" % (tot / 1e3, sum(e.count for e in ka)))
for e in sorted(ka, key=lambda e: -e.self_cpu_time_total)[:60]:
    print("%8.1f us self  %8.1f us total  x%4d  (%.1f us/call)  %s" % (e.self_cpu_time_total, e.cpu_time_total, e.count, e.self_cpu_time_total / e.count, e.key[:70]))
