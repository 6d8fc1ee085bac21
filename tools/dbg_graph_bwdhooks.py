"""Diagnostic: tiny configuration, whole-step graph: module backward hooks stash a copy of every module's grad_output inside
the capture; after each replay the copies are compared with replay 0 in backward execution order."""
import os, sys
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
stash = []
fstash = []
ACTIVE = [False]
DEPTH = int(os.environ.get("DEPTH", "4"))
def hook(name):
    def fwd(mod, inp, out):
        if not (ACTIVE[0] and torch.cuda.is_current_stream_capturing()):
            return
        ts = []
        torch.utils._pytree.tree_map_only(torch.Tensor, ts.append, out)
        for a in (vars(out).values() if hasattr(out, "__dict__") and not torch.is_tensor(out) else []):
            torch.utils._pytree.tree_map_only(torch.Tensor, ts.append, a)
        for i, t in enumerate(ts):
            if t.is_floating_point():
                fstash.append(("%s#%d" % (name, i), t.detach().clone()))
            if t.requires_grad and t.is_floating_point():
                def th(g, i=i):
                    stash.append(("%s#%d" % (name, i), [g.detach().clone()], []))
                t.register_hook(th)
    return fwd
for name, m in model.named_modules():
    if name and name.count(".") < DEPTH:
        m.register_forward_hook(hook(name))
crit.iter = 0
opt = bench.make_optimizer(model, args, fused=False)
ACTIVE[0] = True
step = bench.GraphStep(model, crit, opt, make_samples, text, targets, args, None, 1)
print("stashed modules:", len(stash), "graph:", step.graph_stats)
ref = None
for rep in range(7):
    step.graph.replay(); torch.cuda.synchronize()
    cur = [(n, [t.clone() for t in go], [t.clone() for t in gi]) for n, go, gi in stash]
    fcur = [(n, t.clone()) for n, t in fstash]
    if ref is None:
        ref = cur; fref = fcur; continue
    fd = ["%d:%s %d elems" % (i, n, int((a != b).sum())) for i, ((n, a), (_, b)) in enumerate(zip(fcur, fref)) if not torch.equal(a, b)]
    print("replay", rep, "FORWARD outputs not bitwise equal to replay 0:", len(fd), fd[:6])
    diffs = []
    for i, ((n, go, gi), (_, go0, gi0)) in enumerate(zip(cur, ref)):
        do = max([((a - b).abs().max().item() / (b.abs().max().item() + 1e-20)) for a, b in zip(go, go0)] + [0])
        di = max([((a - b).abs().max().item() / (b.abs().max().item() + 1e-20)) for a, b in zip(gi, gi0)] + [0])
        if do > 1e-3 or di > 1e-3:
            diffs.append("%d:%s out %.1e in %.1e" % (i, n, do, di))
    print("replay", rep, "first modules (backward order) whose grad_output / grad_input differ from replay 0:", diffs[:10])
# detail of the first differing entry of the last replay
for i, ((n, go, gi), (_, go0, gi0)) in enumerate(zip(cur, ref)):
    if not go: continue
    a, b = go[0], go0[0]
    d = (a - b).abs()
    if d.max().item() > 1e-3 * b.abs().max().item():
        nz = (d > 1e-4 * b.abs().max()).nonzero()
        print("detail", i, n, "shape", tuple(a.shape), "differing elements:", nz.shape[0], "of", a.numel(), "max|b|", b.abs().max().item())
        for idx in nz[:12].tolist():
            print("    ", idx, "replay0 %.6g  now %.6g" % (b[tuple(idx)].item(), a[tuple(idx)].item()))
        break
