import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")): sys.path.insert(0, p)
import torch, cases, model_checks
from conftest import Golden
from ocpg_amd.models import amp_cache
from ocpg_amd.util.misc import NestedTensor
dev = torch.device("cuda:0"); g = Golden("e2e_tiny"); meta = g.meta
def run(enabled, amp=True):
    amp_cache.ENABLED = enabled
    torch.manual_seed(0)
    args, model, crit = model_checks.build_product(meta, dev)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta["nopad_sizes"], dev)
    model.train(), crit.train()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        out = model(NestedTensor(x, mask), model_checks.text_for(B, dev), targets)
        losses, *_ = crit(out, targets)
        total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    gn = {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None}
    return out["pred_masks"].detach().float().cpu(), total.item(), gn, {k: v.item() for k, v in losses.items()}
a, b, c = run(False), run(False), run(True)
rel = lambda u, v: float((u - v).norm() / (v.norm() + 1e-12))
print("masks off/off", rel(b[0], a[0]), "on/off", rel(c[0], a[0]), "total", a[1], b[1], c[1])
for k in a[3]:
    print("  loss %-28s %.5f %.5f %.5f" % (k, a[3][k], b[3][k], c[3][k]))
rows = []
for k in a[2]:
    na, nb, nc = a[2][k].norm().item(), b[2][k].norm().item(), c[2][k].norm().item()
    rows.append((abs(nc - na) / (na + 1e-12), k, na, nb, nc, rel(b[2][k], a[2][k]), rel(c[2][k], a[2][k])))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%.4f %-70s off %.5g off2 %.5g on %.5g  elemrel off/off %.4f on/off %.4f" % r)
print("median normdiff", rows[len(rows)//2][0])

print("---- fp32, two identical runs")
a, b = run(True, False), run(True, False)
print("masks", rel(b[0], a[0]), "total", a[1], b[1])
rows = sorted(((rel(b[2][k], a[2][k]), k) for k in a[2]), reverse=True)
for r in rows[:12]: print("%.3e %s" % r)
print("median", rows[len(rows)//2][0])
