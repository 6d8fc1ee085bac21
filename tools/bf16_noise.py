"""Diagnostic for VERDICT r1 'weak #1': where does run-to-run noise enter the bf16 bench mode, and how far is that mode from
the fp32 reference vectors?  Runs the e2e fixture's step twice in bench mode (bf16 autocast + channels-last + amp_cache) from
identical state and reports (1) the first forward module whose output differs between the two runs, (2) the per-parameter
gradient differences run-vs-run, (3) bench mode vs the fp32 reference fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import cases, model_checks
from conftest import Golden
from ocpg_amd.util.misc import NestedTensor

dev = torch.device("cuda:0")
if os.environ.get("DETERMINISTIC") == "1":
    torch.backends.cudnn.deterministic = True
    print("torch.backends.cudnn.deterministic = True")
name, tag = os.environ.get("FIXTURE", "e2e_d32"), os.environ.get("TAG", "pad")
g = Golden(name)
meta = g.meta
B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]


def run(record):
    args, model, crit = model_checks.build_product(meta, dev)
    model_checks.to_channels_last(model)
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"], dev)
    model.train(), crit.train()
    names = {m: n for n, m in model.named_modules()}
    hooks = []
    def hook(m, i, o):
        t = o[0] if isinstance(o, (tuple, list)) and len(o) and torch.is_tensor(o[0]) else o
        if torch.is_tensor(t):
            record.append((names[m], t.detach().float().clone()))
    for m in model.modules():
        if len(list(m.children())) == 0:
            hooks.append(m.register_forward_hook(hook))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(NestedTensor(x, mask), model_checks.text_for(B, dev), targets)
        losses, *_ = crit(out, targets)
        total = crit.weighted_sum(losses)
    total.backward()
    grads = {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None}
    idx = torch.cat([i[0] for i in out["main_matcher_index"]]).cpu()
    return out, {k: float(v.detach()) for k, v in losses.items()}, float(total.detach()), grads, idx


ra, rb = [], []
oa, la, ta, ga, ia = run(ra)
ob, lb, tb, gb, ib = run(rb)
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
print("== forward, run A vs run B (first 12 leaf modules whose outputs differ)")
n = 0
for (na, a), (nb, b) in zip(ra, rb):
    if a.shape == b.shape and not torch.equal(a, b):
        print(f"   {na:60s} rel diff {rel(a, b):.3e}")
        n += 1
        if n >= 12:
            break
print("   (none)" if n == 0 else "")
print("== matcher indices A/B/ref:", ia.tolist(), ib.tolist(), g[f"{tag}_main_idx"].flatten().tolist())
print("== total A/B/ref:", ta, tb, g[f"{tag}_total"].item())
pm_ref = g[f"{tag}_pred_masks"]
print("== mask logits: A vs B rel %.3e ; A vs fp32 reference rel %.3e, max abs %.3e (|ref| max %.1f)" % (
    rel(oa["pred_masks"].float().cpu(), ob["pred_masks"].float().cpu()), rel(oa["pred_masks"].float().cpu(), pm_ref),
    (oa["pred_masks"].float().cpu() - pm_ref).abs().max(), pm_ref.abs().max()))
d = sorted(((rel(ga[k], gb[k]), k, float(gb[k].norm())) for k in ga if gb[k].norm() > 0), reverse=True)
print("== gradients run A vs run B: median rel %.3e; worst 15:" % d[len(d) // 2][0])
for r, k, nrm in d[:15]:
    print(f"   {k:70s} rel {r:.3e}  |g| {nrm:.3e}")
gn = meta[f"{tag}_grad_norms"]
e = sorted(((abs(float(ga[k].norm()) - v) / (abs(v) + 1e-20), k, v) for k, v in gn.items() if v and k in ga), reverse=True)
print("== gradient NORMS bench mode vs fp32 reference: median rel %.3e; worst 10:" % e[len(e) // 2][0])
for r, k, v in e[:10]:
    print(f"   {k:70s} rel {r:.3e}  |g|ref {v:.3e}")
print("== losses (bench / ref), worst 6:")
lr_ = meta[f"{tag}_losses"]
for r, k in sorted(((abs(la[k] - v) / (abs(v) + 1e-12), k) for k, v in lr_.items()), reverse=True)[:6]:
    print(f"   {k:20s} {la[k]:.5f} / {lr_[k]:.5f}  rel {r:.2e}")
