"""In-process interleaved A/B of a module-level boolean switch on the bench step (same box, same process, same clocks).
usage: ab_inproc.py ocpg_amd.models.backbone:FUSED_CONV_BN [rounds] [steps_per_chunk]
Reports min / median ms per step for switch=False and switch=True."""
import importlib, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from ocpg_amd.models import build_model
spec = sys.argv[1]; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6; chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 8
modname, attr = spec.split(":")
mod = importlib.import_module(modname)
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
for v in (False, True, False, True):
    setattr(mod, attr, v)
    for _ in range(3): step()
torch.cuda.synchronize()
res = {False: [], True: []}
for r in range(rounds):
    for v in (False, True) if r % 2 == 0 else (True, False):
        setattr(mod, attr, v)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(chunk): step()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / chunk * 1e3)
for v in (False, True):
    print("%s=%s: min %.2f  median %.2f  ms/step  (%s)" % (attr, v, min(res[v]), statistics.median(res[v]), " ".join("%.1f" % x for x in res[v])))
