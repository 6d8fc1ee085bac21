import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from ocpg_amd.models import build_model
from ocpg_amd.util.misc import NestedTensor
dev = torch.device("cuda:0")
torch.manual_seed(0)
args = bench.model_args(dev, os.environ.get("BB", "resnet50"), amp=True)
model, crit, _ = build_model(args)
model.to(dev).to(memory_format=torch.channels_last); crit.to(dev); model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(int(os.environ.get("CLIPS", "1")), dev, 42)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
keep = {}
def fb(x, mask, nb):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(NestedTensor(x, mask), text, targets)
        out["num_boxes"] = nb
        ld, *_ = crit(out, targets)
        loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    loss.backward()
    ex = (sys.argv[2] if len(sys.argv) > 2 else "")
    which = os.environ.get("KEEP", "all")
    if "k" not in ex:
        if which in ("all", "out"):
            keep.update({k: v for k, v in out.items() if isinstance(v, torch.Tensor)})
        if which in ("all", "ld"):
            keep.update(ld)
        if which.startswith("out:"):
            keep.update({k: v for k, v in out.items() if isinstance(v, torch.Tensor) and k in which[4:].split(",")})
    return loss.detach()
first = make_samples(); x, mask = first.tensors.clone(), first.mask.clone()
nb = crit.global_num_boxes(targets, dev).clone()
crit.iter_device = torch.zeros((), device=dev)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        opt.zero_grad(set_to_none=True); fb(x.clone(), mask.clone(), nb)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = fb(x.clone(), mask.clone(), nb)
params = [p for p in model.parameters() if p.requires_grad]
extra = sys.argv[2] if len(sys.argv) > 2 else ""
for i in range(int(os.environ.get("ITERS", "5"))):
    if "x" in extra:
        s_ = make_samples(); x.copy_(s_.tensors); mask.copy_(s_.mask)
    if "n" in extra:
        nb.copy_(crit.global_num_boxes(targets, dev))
    g.replay()
    if "i" in extra:
        crit.iter_device += 4
    if "s" not in extra:
        torch.cuda.synchronize()
    badk = [k for k, v in keep.items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
    badg = sum(1 for p in params if not torch.isfinite(p.grad).all())
    gn = torch.norm(torch.stack([p.grad.float().norm() for p in params]))
    print(i, "loss", float(loss), "bad outs", badk[:6], "bad grads", badg, "gnorm", float(gn))
    if mode == "full":
        torch.nn.utils.clip_grad_norm_(params, 0.1, foreach=True)
        opt.step()
    elif mode == "noclip":
        opt.step()
    badp = sum(1 for p in model.parameters() if not torch.isfinite(p).all())
    print("   bad params after step", badp)
