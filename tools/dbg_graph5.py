"""Bisect the HIP-graph replay hazard: capture one sub-network (fwd+bwd) at a time, perturb its parameters between
replays, and compare every replay with an eager evaluation at the same parameters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from ocpg_amd.models import build_model
from ocpg_amd.util.misc import NestedTensor
dev = torch.device("cuda:0")
torch.manual_seed(0)
args = bench.model_args(dev, "resnet50", amp=True); args.dropout = 0.0
model, crit, _ = build_model(args)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
model.to(dev).train()
which = sys.argv[1]
x = torch.randn(5, 3, 384, 640, device=dev)
mask = torch.zeros(5, 384, 640, dtype=torch.bool, device=dev)
if which == "backbone":
    net = model.backbone[0]
    terms = {}
    def fn():
        out = net(NestedTensor(x, mask))
        ms = [o.tensors.float().square().mean() for o in out.values()]
        t1 = ms[0] + 0; t2 = t1 + ms[1]; t3 = t2 + ms[2]; t4 = t3 + ms[3]
        terms.update(ms=ms, parts=[t1, t2, t3, t4])
        return t4
elif which == "encoder":
    tr = model.transformer
    srcs = [torch.randn(5, 256, h, w, device=dev) for h, w in ((48, 80), (24, 40), (12, 20), (6, 10))]
    masks = [torch.zeros(5, h, w, dtype=torch.bool, device=dev) for h, w in ((48, 80), (24, 40), (12, 20), (6, 10))]
    tgt = torch.randn(1, 5, 5, 256, device=dev); qe = torch.randn(5, 256, device=dev)
    net = tr
    def fn():
        hs, mem, *_ = tr(srcs, tgt, masks, srcs, qe)
        return hs.float().square().mean() + sum(m.float().square().mean() for m in mem)
elif which == "lfm":
    net = model.input_fft[0]
    s0 = torch.randn(5, 256, 48, 80, device=dev)
    def fn():
        y, g = net(s0)
        return y.float().square().mean()
params = [p for p in net.parameters() if p.requires_grad]
def eager():
    for p in params: p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l = fn()
    l.backward()
    return float(l), float(torch.norm(torch.stack([p.grad.float().norm() for p in params])))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): eager()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
for p in params: p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = fn()
    loss.backward()
grads = [p.grad for p in params]
for i in range(5):
    g.replay(); torch.cuda.synchronize()
    early = float(loss)
    import time; time.sleep(float(os.environ.get('SLEEP', '0')))
    torch.cuda.synchronize()
    if which == 'backbone': print('   ms', [round(float(v), 4) for v in terms['ms']], 'parts', [round(float(v), 4) for v in terms['parts']], 'ptrs', [hex(v.data_ptr()) for v in terms['parts']], hex(loss.data_ptr()))
    gl, gg = float(loss), float(torch.norm(torch.stack([q.float().norm() for q in grads])))
    saved = [q.clone() for q in grads]
    el, eg = eager()
    for p, q in zip(params, saved): p.grad = q          # restore the graph's static grad tensors
    for p, q in zip(params, grads): p.grad = q
    print(which, i, "graph loss %.6f gnorm %.5f | eager loss %.6f gnorm %.5f" % (gl, gg, el, eg), flush=True)
    with torch.no_grad():
        for p in params: p.add_(torch.randn_like(p) * 1e-3 * p.abs().mean())
