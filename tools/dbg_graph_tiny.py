"""Diagnostic: tiny end-to-end configuration, fp32: run-to-run gradient noise of the eager step vs the difference between a
graph replay and the eager step (same parameters)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch, bench, cases, model_checks
from conftest import Golden
from ocpg_amd.util.misc import NestedTensor
dev = torch.device("cuda:0")
meta = Golden("e2e_tiny").meta
args, model, crit = model_checks.build_product(meta, dev)
model_checks.to_channels_last(model)
model.train(); crit.train()
T, H, W = meta["T"], meta["H"], meta["W"]
x, mask, targets = cases.e2e_inputs(2, T, H, W, meta["nopad_sizes"], dev)
text = model_checks.text_for(2, dev)
make_samples = lambda: NestedTensor(x.clone(), mask.clone())
def grads():
    crit.iter = 0
    model.zero_grad(set_to_none=True)
    l = bench.forward_backward(model, crit, make_samples(), text, targets, None)
    return float(l), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
l1, g1 = grads(); l2, g2 = grads()
model.zero_grad(set_to_none=True)
crit.iter = 0
opt = bench.make_optimizer(model, args, fused=False)
import traceback, collections
from torch.utils._python_dispatch import TorchDispatchMode
class H2D(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.hits = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if torch.cuda.is_current_stream_capturing():
            ins = []
            torch.utils._pytree.tree_map_only(torch.Tensor, ins.append, (args, kwargs))
            outs = []
            torch.utils._pytree.tree_map_only(torch.Tensor, outs.append, out)
            if any(t.device.type == "cpu" for t in ins) and any(t.is_cuda for t in outs + ins):
                fr = [f for f in traceback.extract_stack() if "ocpg_amd" in f.filename or "bench.py" in f.filename]
                where = "%s:%d" % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else "?"
                pinned = [t.is_pinned() for t in ins if t.device.type == "cpu"]
                self.hits[(str(func), where, str(pinned))] += 1
        return out
h2d = H2D()
with h2d:
    step = bench.GraphStep(model, crit, opt, make_samples, text, targets, args, None, 1)
print("host->device ops inside the capture:")
for k, v in h2d.hits.items(): print("   ", v, k)
print("graph:", step.graph_stats)
import collections
print("memcpy nodes (kind, bytes) -> count:", dict(collections.Counter((k, b) for k, b, _, _ in step.memcpy_nodes)))
for rep in range(2):
    step.graph.replay(); torch.cuda.synchronize()
    gg = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    rows = []
    for k in g1:
        m = g1[k].abs().max().item() + 1e-12
        rows.append(((gg[k] - g1[k]).abs().max().item() / m, (g2[k] - g1[k]).abs().max().item() / m, k))
    rows.sort(reverse=True)
    print("replay", rep, "loss graph %.6f eager %.6f %.6f" % (float(step.loss), l1, l2))
    for r in rows[:8]:
        print("   graph-vs-eager %.2e   eager-vs-eager %.2e   %s" % r)
print("---- replay-to-replay")
ref = None
for rep in range(6):
    step.graph.replay(); torch.cuda.synchronize()
    gg = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    if ref is None:
        ref = gg
        continue
    rows = sorted(((gg[k] - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 1e-12), (gg[k] - g1[k]).abs().max().item() / (g1[k].abs().max().item() + 1e-12), k) for k in ref)[::-1]
    print("replay", rep, "vs replay 0 / vs eager:", ["%.1e/%.1e %s" % (a, b, k[-40:]) for a, b, k in rows[:4]])
