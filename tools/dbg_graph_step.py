"""Diagnostic for the whole-step HIP graph (bench.py --graph): capture forward+criterion+backward, replay it with the optimizer
step in between and compare every replay's loss / gradient norm with an eager evaluation at the same parameters.
Run as `python -X faulthandler tools/dbg_graph_step.py [n_replays]`; env KEEP=0 drops the keep-every-tensor mode,
SYNC=0 drops the host fences around the replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from ocpg_amd.models import build_model

def say(*a):
    print(*a, flush=True)

dev = torch.device("cuda:0")
if os.environ.get("OCPG_BLAS"):
    torch.backends.cuda.preferred_blas_library(os.environ["OCPG_BLAS"])
from ocpg_amd.models.amp_cache import lookup as amp_lookup
if os.environ.get("MMFFT") == "1":       # experiment: dense-DFT matmuls instead of rocFFT
    import math
    _F = {}
    def _dft(n, inverse):
        k = (n, inverse)
        if k not in _F:
            idx = torch.arange(n, device=dev, dtype=torch.float64)
            ang = 2 * math.pi * torch.outer(idx, idx) / n * (1 if inverse else -1)
            _F[k] = torch.complex(torch.cos(ang), torch.sin(ang)).to(torch.complex64)
        return _F[k]
    def _fft2(x, s=None, **kw):
        h, w = x.shape[-2:]
        return _dft(h, False) @ x.to(torch.complex64) @ _dft(w, False)
    def _ifft2(x, s=None, **kw):
        h, w = x.shape[-2:]
        return (_dft(h, True) @ x.to(torch.complex64) @ _dft(w, True)) / (h * w)
    torch.fft.fft2, torch.fft.ifft2 = _fft2, _ifft2
if os.environ.get("SMALLK") == "0":
    import ocpg_amd.models.attention as _att
    _att.HIP_SMALLK = False
torch.manual_seed(0)
args = bench.model_args(dev, os.environ.get("BACKBONE", "resnet101"), amp=True)
args.dropout = 0.0
model, crit, _ = build_model(args)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if hasattr(m, "dropout_p"):
        m.dropout_p = 0.0
model.to(dev); crit.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(int(os.environ.get("CLIPS", "2")), dev, 42)
params = [p for p in model.parameters() if p.requires_grad]

def gnorm(grads):
    return float(torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float()) for g in grads])))

def eager(it):
    crit.iter = it
    saved_dev = crit.iter_device
    crit.iter_device = None
    for p in params: p.grad = None
    l = float(bench.forward_backward(model, crit, make_samples(), text, targets, torch.bfloat16))
    g = gnorm([p.grad for p in params])
    crit.iter_device = saved_dev
    return l, g

say("eager 0:", eager(0))
if os.environ.get("KEEP", "1") == "0":
    class _NoKeep:
        keep = []
        def __enter__(self): return self
        def __exit__(self, *a): return False
    bench._KeepEveryTensor = _NoKeep
STAGE = os.environ.get("STAGE")
if STAGE:
    # capture only a prefix of the step: fwd | crit | bwd
    from ocpg_amd.util.misc import NestedTensor
    first = make_samples()
    x, mask = first.tensors.clone(), first.mask.clone()
    nb = crit.global_num_boxes(targets, dev).clone()
    crit.iter_device = torch.zeros((), device=dev)
    def part():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(NestedTensor(x.clone(), mask.clone()), text, targets)
            if STAGE == "fwd":
                return out, None
            out["num_boxes"] = nb
            ld, *_ = crit(out, targets)
            loss = crit.weighted_sum(ld)
        if STAGE == "bwd":
            loss.backward()
        return out, loss
    import gc; crit._last = None; gc.collect()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for p in params: p.grad = None
            part()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    for p in params: p.grad = None
    g = torch.cuda.CUDAGraph()
    say("capturing stage", STAGE)
    stash = []
    DEPTH = int(os.environ.get("DEPTH", "2"))
    def hook(name):
        def f(mod, inp, out):
            ts = []
            torch.utils._pytree.tree_map_only(torch.Tensor, ts.append, out)
            for a in vars(out).values() if hasattr(out, "__dict__") and not torch.is_tensor(out) else []:
                torch.utils._pytree.tree_map_only(torch.Tensor, ts.append, a)
            if os.environ.get("HOOK_IN") and name.startswith(os.environ["HOOK_IN"]):
                stash.append((name + ":in", [t for t in inp if torch.is_tensor(t) and t.is_floating_point()]))
                stash.append((name + ":w", [amp_lookup(p_) for p_ in mod.parameters(recurse=False)]))
            stash.append((name, [t for t in ts if t.is_floating_point()]))
        return f
    for name, m in model.named_modules():
        if name and name.count(".") < DEPTH:
            m.register_forward_hook(hook(name))
    with torch.cuda.graph(g, stream=side):
        res = part()
    say("captured stage", STAGE, "hooked outputs", len(stash))
    ref = None
    for r in range(3):
        g.replay(); torch.cuda.synchronize()
        sums = [(name, [float(t.float().abs().sum()) for t in ts]) for name, ts in stash]
        bad = [(i, name) for i, (name, v) in enumerate(sums) if any(x != x or x == float("inf") for x in v)]
        say("replay", r, "loss", None if res[1] is None else float(res[1]), "first non-finite:", bad[:3])
        if ref is None:
            ref = sums
        else:
            diff = [(i, a[0], a[1][:2], b[1][:2]) for i, (a, b) in enumerate(zip(ref, sums)) if any(abs(x - y) > 2e-2 * abs(x) + 1e-6 or y != y for x, y in zip(a[1], b[1]))]
            say("   first outputs differing from replay 0:", diff[:4])
    sys.exit(0)
say("capturing")
step = bench.GraphStep(model, crit, opt, make_samples, text, targets, args, torch.bfloat16, 1)
say("captured; memset nodes replaced:", step.memset_nodes_replaced)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
NO_EAGER, NO_OPT = os.environ.get("NO_EAGER") == "1", os.environ.get("NO_OPT") == "1"
for i in range(n):
    s = make_samples()
    step.x.copy_(s.tensors); step.mask.copy_(s.mask)
    if os.environ.get("SYNC", "1") != "0":
        torch.cuda.synchronize()
    step.graph.replay()
    if os.environ.get("SYNC", "1") != "0":
        torch.cuda.synchronize()
    gl, gg = float(step.loss), gnorm(step.grads)
    it = int(step.criterion.iter_device.item())
    bad_terms = [k for k, v in step.static["loss_dict"].items() if not bool(torch.isfinite(v).all())]
    bad_out = [k for k, v in step.static["out"].items() if torch.is_tensor(v) and v.is_floating_point() and not bool(torch.isfinite(v).all())]
    bad_grads = sum(1 for g in step.grads if not bool(torch.isfinite(g).all()))
    if NO_EAGER:
        el = eg = float("nan")
    else:
        keep = [g.clone() for g in step.grads]
        el, eg = eager(it)
        for p, g, k in zip(params, step.grads, keep):
            g.copy_(k); p.grad = g
    say(f"replay {i}: graph loss {gl:.5f} gnorm {gg:.4f} | eager loss {el:.5f} gnorm {eg:.4f}  (criterion iter {it}) bad terms {bad_terms[:6]} bad out {bad_out} bad grads {bad_grads}/{len(step.grads)}")
    step.criterion.iter_device += step.calls_per_fwd
    if not NO_OPT:
        torch.nn.utils.clip_grad_norm_(params, args.clip_max_norm, foreach=True)
        opt.step()
say("done")
