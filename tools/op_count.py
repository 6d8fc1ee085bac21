"""Diagnostic: aten ops dispatched per module path in ONE forward+criterion (TorchDispatchMode + module hooks), and the autograd
nodes of the resulting graph per node type.  Tells where the launch count of the step comes from."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from ocpg_amd.models import build_model

dev = torch.device("cuda:0")
args = bench.model_args(dev, "resnet101", amp=True)
model, crit, _ = build_model(args)
model.to(dev); crit.to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.to(memory_format=torch.channels_last)
model.train(); crit.train()
make_samples, text, targets = bench.synthetic_batch(2, dev, 42)
DEPTH = int(os.environ.get("DEPTH", "3"))
stack = ["<top>"]
names = {m: n for n, m in model.named_modules()}
names.update({m: "criterion." + n for n, m in crit.named_modules()})
def pre(m, i): stack.append(".".join(names.get(m, "?").split(".")[:DEPTH]) or "<model>")
def post(m, i, o): stack.pop()
for m in list(model.modules()) + list(crit.modules()):
    m.register_forward_pre_hook(pre); m.register_forward_hook(post)
VIEW = {"view", "_unsafe_view", "reshape", "permute", "transpose", "t", "expand", "unsqueeze", "squeeze", "select", "slice", "detach", "alias",
        "as_strided", "unbind", "split", "split_with_sizes", "unflatten", "flatten", "view_as_real", "view_as_complex", "chunk", "unfold",
        "lift_fresh", "_reshape_alias", "empty", "empty_like", "empty_strided", "sym_size", "stride", "size", "numel", "is_same_size", "prim"}
by_fn = collections.Counter(); by_line = collections.Counter(); copies = collections.Counter(); copy_elems = collections.Counter()
counts = collections.Counter(); per_op = collections.defaultdict(collections.Counter)
class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEW:
            counts[stack[-1]] += 1; per_op[stack[-1]][name] += 1
            f = sys._getframe(1)
            while f is not None and not ("ocpg_amd" in f.f_code.co_filename or f.f_code.co_filename.endswith("bench.py")):
                f = f.f_back
            if f is not None:
                by_fn["%s:%s" % (os.path.basename(f.f_code.co_filename), f.f_code.co_name)] += 1
                by_line["%s:%d" % (os.path.basename(f.f_code.co_filename), f.f_lineno)] += 1
                if name in ("clone", "copy_", "_to_copy", "cat", "stack", "fill_", "zeros", "zero_", "index_select", "gather"):
                    big = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
                    copies["%s:%d %s" % (os.path.basename(f.f_code.co_filename), f.f_lineno, name)] += 1
                    copy_elems["%s:%d %s" % (os.path.basename(f.f_code.co_filename), f.f_lineno, name)] += big
        return func(*args, **(kwargs or {}))
def run():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(make_samples(), text, targets)
        ld, *_ = crit(out, targets)
        return sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
run().backward(); model.zero_grad()
with Count():
    loss = run()
print("forward kernels-ish ops per module (views excluded): total", sum(counts.values()))
for k, v in counts.most_common(40):
    print("%5d  %-50s %s" % (v, k, " ".join("%s:%d" % kv for kv in per_op[k].most_common(8))))
print("--- per function"); print("\n".join("%5d %s" % (v, k) for k, v in by_fn.most_common(45)))
print("--- per line"); print("\n".join("%5d %s" % (v, k) for k, v in by_line.most_common(60)))
print("--- copies by line (count, Melem)"); print("\n".join("%4d %8.2f %s" % (copies[k], v / 1e6, k) for k, v in copy_elems.most_common(45)))
# autograd graph census
seen, todo, types = set(), [loss.grad_fn], collections.Counter()
while todo:
    n = todo.pop()
    if n is None or n in seen: continue
    seen.add(n); types[type(n).__name__] += 1
    todo.extend(f for f, _ in n.next_functions)
print("autograd nodes:", len(seen))
print(" ".join("%s:%d" % kv for kv in types.most_common(60)))
