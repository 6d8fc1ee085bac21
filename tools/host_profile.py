"""cProfile of the host side of a few bench steps (where does the Python / dispatch time go?)."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from ocpg_amd.models import build_model
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
args = bench.model_args(dev, "resnet101", amp=True)
model, crit, _ = build_model(args)
model.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d): m.to(memory_format=torch.channels_last)
crit.to(dev); model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(2, dev, 42)
step = bench.EagerStep(model, model, crit, opt, make_samples, text, targets, args, torch.bfloat16)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:6000])
