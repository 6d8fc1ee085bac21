"""Diagnostic: one bench step under torch.profiler; GPU time and launch count per (aten op, innermost ocpg_amd source line).
Backward ops have no Python stack: they are attributed to their autograd node name instead."""
import os, sys, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda:0")
from ocpg_amd.models import build_model
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
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
agg = collections.defaultdict(lambda: [0.0, 0, set()])
for e in ev:
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0 or e.device_type != torch.autograd.DeviceType.CPU:
        continue
    where = "?"
    for fr in (e.stack or []):
        if "ocpg_amd" in fr or "bench.py" in fr:
            m = re.search(r"(ocpg_amd/[\w/]+\.py|bench\.py)\((\d+)\)", fr)
            where = "%s:%s" % (m.group(1), m.group(2)) if m else fr[:60]
            break
    if where == "?":
        # walk up to the enclosing autograd node / parent op name
        p = e.cpu_parent
        while p is not None and not ("Backward" in p.name or "autograd::engine" in p.name):
            p = p.cpu_parent
        where = "bwd:" + (p.name if p is not None else "-")
    k = (e.name, where + (" " + str(e.input_shapes)[:80] if e.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::miopen_convolution", "aten::convolution_backward") else ""))
    agg[k][0] += dt; agg[k][1] += 1
    if e.input_shapes:
        agg[k][2].add(str(e.input_shapes)[:90])
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print("total device time %.2f ms over %d ops" % (tot / 1e3, sum(v[1] for v in agg.values())))
for (name, where), (t, n, shp) in rows[:int(os.environ.get("TOP", "120"))]:
    print("%8.1f us %4d  %-38s %-60s %s" % (t, n, name[:38], where[:60], sorted(shp)[0] if shp else ""))
