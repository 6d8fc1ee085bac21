"""Diagnostic: poison the free blocks of the step graph's private pool with NaN between replays (a second graph sharing the
pool fills many scratch tensors): a kernel that is NOT in the graph (ran only at capture time), or that reads memory no
kernel of this replay wrote, then shows up as NaN."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch, bench, cases, model_checks
from conftest import Golden
from ocpg_amd.util.misc import NestedTensor
dev = torch.device("cuda:0")
AMP = torch.bfloat16 if os.environ.get("AMP") == "1" else None
meta = Golden("e2e_tiny").meta
args, model, crit = model_checks.build_product(meta, dev)
model_checks.to_channels_last(model)
model.train(); crit.train()
T, H, W = meta["T"], meta["H"], meta["W"]
x, mask, targets = cases.e2e_inputs(2, T, H, W, meta["nopad_sizes"], dev)
text = model_checks.text_for(2, dev)
make_samples = lambda: NestedTensor(x.clone(), mask.clone())
crit.iter = 0
opt = bench.make_optimizer(model, args, fused=False)
step = bench.GraphStep(model, crit, opt, make_samples, text, targets, args, AMP, 1)
step.graph.replay(); torch.cuda.synchronize()
ref = [g.clone() for g in step.grads]
print("loss", float(step.loss))
g2 = torch.cuda.CUDAGraph()
keep = []
with torch.cuda.graph(g2, pool=step.graph.pool()):
    for nbytes, cnt in ((512, 1500), (8192, 800), (65536, 500), (1 << 20, 200), (1 << 23, 40), (1 << 26, 6)):
        for _ in range(cnt):
            keep.append(torch.empty(nbytes // 4, device=dev))
    for t_ in keep:
        t_.fill_(float("nan"))
print("poison tensors:", len(keep), "MB:", sum(t.numel() for t in keep) * 4 >> 20)
names = [k for k, p in model.named_parameters() if p.requires_grad]
for r in range(4):
    g2.replay(); torch.cuda.synchronize()
    step.graph.replay(); torch.cuda.synchronize()
    bad = [n for n, g in zip(names, step.grads) if not bool(torch.isfinite(g).all())]
    d = max(((g - g0).abs().max() / (g0.abs().max() + 1e-20)).item() for g, g0 in zip(step.grads, ref) if bool(torch.isfinite(g).all()))
    print("replay", r, "loss", float(step.loss), "NaN grads:", len(bad), bad[:6], "max rel diff of finite grads vs first replay %.2e" % d, flush=True)
    bad_terms = [k for k, v in step.static["loss_dict"].items() if not bool(torch.isfinite(v).all())]
    print("    NaN loss terms:", bad_terms[:8])
