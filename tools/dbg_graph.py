import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from ocpg_amd.models import build_model
dev = torch.device("cuda:0")
drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
torch.manual_seed(0)
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "0") == "1"
args = bench.model_args(dev, os.environ.get("BB", "resnet50"), amp=True); args.dropout = drop
model, crit, _ = build_model(args)
if drop == 0:
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
model.to(dev).to(memory_format=torch.channels_last); crit.to(dev); model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(int(os.environ.get("CLIPS", "1")), dev, 42)
eager = bench.EagerStep(model, model, crit, opt, make_samples, text, targets, args, torch.bfloat16)
print("eager", [float(eager()) for _ in range(int(os.environ.get("EAGER_STEPS", "3")))])
gs = bench.GraphStep(model, crit, opt, make_samples, text, targets, args, torch.bfloat16, 1)
for i in range(6):
    l = gs()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("graph", i, float(l), "non-finite grads:", len(bad), bad[:5])
