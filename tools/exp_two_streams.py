"""Experiment: two one-clip steps captured as two graphs and replayed on two streams at once vs one after the other (timing only: both
graphs accumulate into the same gradient buffers).  Is there concurrency to be had from running half-batches side by side?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ocpg_amd.models import build_model

dev = torch.device("cuda:0")
args = bench.model_args(dev, "resnet101", amp=True)
torch.manual_seed(42)
model, crit, _ = build_model(args)
model.to(dev), crit.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
model.train(), crit.train()
opt = bench.make_optimizer(model, args)
steps = []
for seed in (42, 43):
    make_samples, text, targets = bench.synthetic_batch(1, dev, seed)
    steps.append(bench.GraphStep(model, crit, opt, make_samples, text, targets, args, torch.bfloat16, 1))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def one():
    steps[0].replay()


def serial():
    steps[0].replay()
    steps[1].replay()


def concurrent():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur), s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        steps[0].replay()
    with torch.cuda.stream(s2):
        steps[1].replay()
    cur.wait_stream(s1), cur.wait_stream(s2)


print("one 1-clip graph      : %.2f ms" % timed(one))
print("two, same stream      : %.2f ms" % timed(serial))
print("two, two streams      : %.2f ms" % timed(concurrent))
