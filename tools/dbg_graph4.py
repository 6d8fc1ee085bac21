import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from ocpg_amd.models import build_model
dev = torch.device("cuda:0")
torch.manual_seed(0)
args = bench.model_args(dev, "resnet50", amp=True)
model, crit, _ = build_model(args)
model.to(dev).to(memory_format=torch.channels_last); crit.to(dev); model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(1, dev, 42)
trace = []
def hook(name):
    def f(m, i, o):
        def flat(x):
            if isinstance(x, torch.Tensor): return [x]
            if isinstance(x, (list, tuple)): return [t for e in x for t in flat(e)]
            if hasattr(x, "tensors"): return [x.tensors]
            return []
        trace.append((name, [t for t in flat(i)], [t for t in flat(o)]))
    return f
for n, m in model.named_modules():
    if n: m.register_forward_hook(hook(n))
for n, m in crit.named_modules():
    m.register_forward_hook(hook("criterion." + n))
gs = bench.GraphStep(model, crit, opt, make_samples, text, targets, args, torch.bfloat16, 1)
names = {id(p): n for n, p in model.named_parameters()}
params = gs.params
def stats(tag):
    torch.cuda.synchronize()
    badg = [names[id(p)] for p in params if not torch.isfinite(p.grad).all()]
    badp = [names[id(p)] for p in params if not torch.isfinite(p).all()]
    gn = torch.norm(torch.stack([p.grad.float().norm() for p in params]))
    print(tag, "loss", float(gs.loss), "gnorm", float(gn), "bad grads", len(badg), badg[:4], "bad params", len(badp), badp[:6], flush=True)
ncap = len(trace)
print("hook records total", ncap)
def first_nan():
    # the LAST forward recorded is the captured one: records of the capture are the tail of `trace`
    per_fwd = ncap // 4            # 3 warm-ups + 1 capture
    for name, ins, outs in trace[-per_fwd:]:
        bi = [tuple(t.shape) for t in ins if t.dtype.is_floating_point and not torch.isfinite(t).all()]
        bo = [tuple(t.shape) for t in outs if t.dtype.is_floating_point and not torch.isfinite(t).all()]
        if bi or bo:
            print("   first non-finite at module", name, "inputs", bi[:3], "outputs", bo[:3])
            return
    print("   no non-finite module outputs")
for i in range(3):
    gs.graph.replay(); stats(f"{i} after replay"); first_nan()
    tn = torch.nn.utils.clip_grad_norm_(params, 0.1, foreach=True); stats(f"{i} after clip (total_norm {float(tn):.3f})")
    opt.step(); stats(f"{i} after step")
    # optimizer state sanity
    bads = 0
    for p in params:
        st = opt.state[p]
        for k, v in st.items():
            if torch.is_tensor(v) and v.dtype.is_floating_point and not torch.isfinite(v).all():
                bads += 1
    print("   bad optimizer state tensors:", bads)
