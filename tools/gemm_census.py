"""Diagnostic: the GEMM calls of one bench step through the plan cache (ocpg_gemm, ocpg_gemm_bn_act), grouped by shape, with HIP-event
time per call -- which shapes carry the hipBLASLt time of the step and how far each is from the matrix-core peak."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ocpg_amd import _lib
from ocpg_amd.models import build_model

dev = torch.device("cuda:0")
args = bench.model_args(dev, "resnet101", amp=True)
model, crit, _ = build_model(args)
model.to(dev), crit.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
model.train(), crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(int(os.environ.get("CLIPS", "2")), dev, 42)
step = bench.EagerStep(model, model, crit, opt, make_samples, text, targets, args, torch.bfloat16)
for _ in range(4):
    step()
torch.cuda.synchronize()
_lib.enable_kernel_timing(True)
n = 3
for _ in range(n):
    step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for name, a, e0, e1 in _lib._TIMING["events"]:
    if name == "ocpg_gemm":
        key = ("gemm", "bf16" if a[4] else "fp32", "T" if a[6] else "N", "T" if a[7] else "N", a[8], a[9], a[10], a[14])
        fl = 2.0 * a[8] * a[9] * a[10] * max(a[14], 1)
    elif name == "ocpg_gemm_bn_act":
        key = ("gemm_bn_act", "bf16" if a[7] else "fp32", "N", "T", a[8], a[9], a[10], 1)
        fl = 2.0 * a[8] * a[9] * a[10]
    else:
        continue
    d = agg[key]
    d[0] += e0.elapsed_time(e1) * 1e3
    d[1] += 1
    d.append(fl) if len(d) == 2 else None
_lib.enable_kernel_timing(False)
tot = sum(v[0] for v in agg.values()) / n
print("total %.2f ms per step in %d calls per step" % (tot / 1e3, sum(v[1] for v in agg.values()) // n))
print("%-12s %-5s %s%s %8s %6s %6s %3s %6s %9s %8s %7s" % ("symbol", "dtype", "A", "B", "M", "N", "K", "b", "calls", "us/call", "us/step", "TFLOPs"))
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    us = v[0] / v[1]
    print("%-12s %-5s %s%s %8d %6d %6d %3d %6.1f %9.1f %8.1f %7.1f" % (*key, v[1] / n, us, v[0] / n, v[2] / us / 1e6))
