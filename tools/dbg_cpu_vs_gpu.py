"""Debug helper: run the product on CPU (MSDA test double) and on the GPU (HIP op) and print per-module max diffs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from conftest import Golden
import model_checks, cases
from ocpg_amd.util.misc import NestedTensor
import ocpg_amd.models.ops.modules.ms_deform_attn as mod
from oracle.msda import MSDAOracleFunction

tag = sys.argv[1] if len(sys.argv) > 1 else "pad"
g = Golden("e2e_tiny"); meta = g.meta
B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
real = mod.MSDeformAttnFunction


def run(device):
    mod.MSDeformAttnFunction = MSDAOracleFunction if device == "cpu" else real
    args, model, crit = model_checks.build_product(meta, torch.device(device))
    store = {}
    def hook(name):
        def f(m, i, o):
            store[name] = o
        return f
    for name, m in model.named_modules():
        if name.count(".") <= 3 and name:
            m.register_forward_hook(hook(name))
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"], device)
    model.train()
    out = model(NestedTensor(x, mask), model_checks.text_for(B, torch.device(device)), targets)
    return store, out


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for e in o for t in flat(e)]
    if hasattr(o, "tensors"):
        return [o.tensors]
    return []

sc, oc = run("cpu")
sg, og = run("cuda")
for k in sc:
    a, b = flat(sc[k]), flat(sg[k])
    d = [(float((p.float() - q.float().cpu()).abs().max()), float(p.float().abs().max())) for p, q in zip(a, b) if p.shape == q.shape and p.dtype.is_floating_point]
    if d and max(x[0] / (x[1] + 1e-9) for x in d) > 2e-5:
        print(f"{k:60s}", " ".join(f"{e:.2e}/{m:.1e}" for e, m in d))
for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low"):
    print(k, float((oc[k] - og[k].cpu()).abs().max()), "vs golden cpu", float((oc[k] - g[f"{tag}_{k}"]).abs().max()), "gpu", float((og[k].cpu() - g[f"{tag}_{k}"]).abs().max()))
