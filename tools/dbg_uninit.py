"""Diagnostic: every torch.empty* result is filled with NaN (float) before use, for one eager forward+backward: a kernel that reads
memory it (or its producer) never wrote turns its consumers NaN.  Lists the loss terms / gradients that became NaN."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch, bench, cases, model_checks
from conftest import Golden
from ocpg_amd.util.misc import NestedTensor
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0")
FULL = os.environ.get("FULL") == "1"
AMP = torch.bfloat16 if os.environ.get("AMP") == "1" else None
if FULL:
    from ocpg_amd.models import build_model
    args = bench.model_args(dev, "resnet101", amp=AMP is not None)
    model, crit, _ = build_model(args); model.to(dev); crit.to(dev)
    make_samples, text, targets = bench.synthetic_batch(2, dev, 42)
else:
    meta = Golden("e2e_tiny").meta
    args, model, crit = model_checks.build_product(meta, dev)
    T, H, W = meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(2, T, H, W, meta["nopad_sizes"], dev)
    text = model_checks.text_for(2, dev)
    make_samples = lambda: NestedTensor(x.clone(), mask.clone())
model_checks.to_channels_last(model)
model.train(); crit.train()
class Poison(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.n = 0; self.seen_threads = set()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if "empty" in name and isinstance(out, torch.Tensor) and out.is_cuda and out.is_floating_point():
            import threading
            self.seen_threads.add(threading.get_ident())
            with torch.no_grad():
                torch.ops.aten.fill_.Scalar(out, float("nan"))
            self.n += 1
        return out
for _ in range(2):
    bench.forward_backward(model, crit, make_samples(), text, targets, AMP); model.zero_grad(set_to_none=True)
crit.iter = 0
po = Poison()
with po:
    with torch.autocast("cuda", dtype=AMP, enabled=AMP is not None):
        out = model(make_samples(), text, targets)
        ld, *_ = crit(out, targets)
        loss = crit.weighted_sum(ld)
    loss.backward()
torch.cuda.synchronize()
print("poisoned allocations:", po.n, "threads:", len(po.seen_threads))
print("NaN loss terms:", [k for k, v in ld.items() if not bool(torch.isfinite(v).all())])
print("NaN outputs:", [k for k, v in out.items() if torch.is_tensor(v) and v.is_floating_point() and not bool(torch.isfinite(v).all())])
bad = [k for k, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
print("NaN gradients: %d of %d" % (len(bad), sum(1 for p in model.parameters() if p.grad is not None)))
for k in bad[:40]: print("   ", k)
