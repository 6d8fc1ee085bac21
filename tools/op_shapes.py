"""Diagnostic: ATen ops of one eager bench step by (op, input shapes), sorted by device time -- which adds / copies / casts / reductions
move the big maps."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from ocpg_amd.models import build_model

dev = torch.device("cuda:0")
args = bench.model_args(dev, os.environ.get("BACKBONE", "resnet101"), amp=True)
model, crit, _ = build_model(args)
model.to(dev), crit.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
model.train(), crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(2, dev, 42)
step = bench.EagerStep(model, model, crit, opt, make_samples, text, targets, args, torch.bfloat16)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
want = tuple(os.environ.get("OPS", "aten::add,aten::add_,aten::copy_,aten::sum,aten::mul,aten::_to_copy,aten::fill_,aten::cat,aten::div,aten::clone").split(","))
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or e.name not in want:
        continue
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0:
        continue
    shapes = str([s for s in (e.input_shapes or []) if s])[:90]
    k = (e.name, shapes)
    agg[k][0] += dt
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print("selected ops: %.2f ms device time in %d calls" % (tot / 1e3, sum(v[1] for v in agg.values())))
for (name, shapes), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print("%8.1f us %4d  %-14s %s" % (t, n, name, shapes))
