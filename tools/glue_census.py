"""Diagnostic: one bench step under torch.profiler with a record_function label per module; device time and launch count per
(module, aten op), backward ops attributed to the module of their forward op (sequence numbers).  Tells where the
elementwise / copy / reduce glue of the step comes from."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from ocpg_amd.models import build_model

dev = torch.device("cuda:0")
# CONFIG5=1: BASELINE config #5 (Video-Swin-B, 8 x 480 x 854, fp16 + GradScaler, 1 clip; text features precomputed)
C5 = os.environ.get("CONFIG5") == "1"
if C5:
    bench.T_FRAMES, bench.HEIGHT, bench.WIDTH = 8, 480, 854
    os.environ.setdefault("BACKBONE", "video_swin_b_p4w7")
AMP = torch.float16 if C5 else torch.bfloat16
args = bench.model_args(dev, os.environ.get("BACKBONE", "resnet101"), amp=True)
model, crit, _ = build_model(args)
model.to(dev); crit.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(1 if C5 else 2, dev, 42)
step = bench.EagerStep(model, model, crit, opt, make_samples, text, targets, args, AMP)
for _ in range(4):
    step()
torch.cuda.synchronize()
DEPTH = int(os.environ.get("DEPTH", "3"))
names = {m: n for n, m in model.named_modules()}
names.update({m: "criterion." + n for n, m in crit.named_modules()})
ctx = {}
def short(n):
    parts = n.split(".")
    # collapse layer indices of the ResNet body so that the 33 bottlenecks aggregate
    return ".".join(parts[:DEPTH])
def pre(m, i):
    r = torch.autograd.profiler.record_function("mod:" + short(names.get(m, "?")))
    r.__enter__(); ctx.setdefault(m, []).append(r)
def post(m, i, o):
    ctx[m].pop().__exit__(None, None, None)
for m in list(model.modules()) + list(crit.modules()):
    m.register_forward_pre_hook(pre); m.register_forward_hook(post)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=os.environ.get("SHAPES") == "1") as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
def mod_of(e):
    p = e
    while p is not None:
        if p.name.startswith("mod:"):
            return p.name[4:]
        p = p.cpu_parent
    return None
fwd_mod = {}
for e in ev:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and "Backward" not in e.name and "autograd::engine" not in e.name:
        m = mod_of(e)
        if m is not None and e.sequence_nr not in fwd_mod:
            fwd_mod[e.sequence_nr] = m
agg = collections.defaultdict(lambda: [0.0, 0])
for e in ev:
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0:
        continue
    m = mod_of(e)
    if m is not None:
        label = "fwd " + m
    else:
        p, node = e, None
        while p is not None:
            if p.name.startswith("autograd::engine::evaluate_function"):
                node = p
                break
            p = p.cpu_parent
        if node is not None:
            label = "bwd " + fwd_mod.get(node.sequence_nr, "?") + " [" + node.name.split(": ")[-1] + "]"
        else:
            label = "top (optimizer / clip / glue)"
    k = (label, e.name)
    agg[k][0] += dt; agg[k][1] += 1
tot_t = sum(v[0] for v in agg.values()); tot_n = sum(v[1] for v in agg.values())
print("total device time %.2f ms over %d ops with kernels" % (tot_t / 1e3, tot_n))
bymod = collections.defaultdict(lambda: [0.0, 0])
for (label, name), (t, n) in agg.items():
    bymod[label.split(" [")[0]][0] += t; bymod[label.split(" [")[0]][1] += n
print("---- per module (fwd / bwd)")
for k, (t, n) in sorted(bymod.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%8.1f us %5d  %s" % (t, n, k))
print("---- per (module, op), by launches")
for (label, name), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "150"))]:
    print("%8.1f us %5d  %-32s %s" % (t, n, name[:32], label[:110]))
print("---- ATen ops only, by device time")
aten = {k: v for k, v in agg.items() if k[1].startswith("aten::")}
print("ATen total %.2f ms over %d ops" % (sum(v[0] for v in aten.values()) / 1e3, sum(v[1] for v in aten.values())))
for (label, name), (t, n) in sorted(aten.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", "150"))]:
    print("%8.1f us %5d  %-28s %s" % (t, n, name[:28], label[:120]))
byop = collections.defaultdict(lambda: [0.0, 0])
for (label, name), (t, n) in aten.items():
    byop[name][0] += t; byop[name][1] += n
print("---- ATen by op")
for k, (t, n) in sorted(byop.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.1f us %5d  %s" % (t, n, k))
if os.environ.get("SHAPES") == "1":
    # ops that move few bytes per microsecond (strided / broadcast / tiny-row access patterns): candidates for a layout fix
    print("---- slow ATen ops (>= 25 us, < 1.5 TB/s of operand bytes assuming fp32)")
    slow = []
    for e in ev:
        dt = getattr(e, "self_device_time_total", 0) or 0
        if dt < 25 or not e.name.startswith("aten::") or not getattr(e, "input_shapes", None):
            continue
        numel = 0
        for sh in e.input_shapes:
            if sh:
                n = 1
                for d in sh:
                    n *= d
                numel += n
        gbs = numel * 4 / dt / 1e3
        if gbs < 1500:
            m = mod_of(e) or "?"
            slow.append((dt, e.name, m, [list(sh) for sh in e.input_shapes if sh][:3], gbs))
    for dt, name, m, shapes, gbs in sorted(slow, key=lambda t: -t[0])[:40]:
        print("%7.1f us  %-24s %-40s %6.0f GB/s  %s" % (dt, name[:24], m[:40], gbs, shapes))
