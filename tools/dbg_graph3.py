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
gs = bench.GraphStep(model, crit, opt, make_samples, text, targets, args, torch.bfloat16, 1)
variant = sys.argv[1]
params = gs.params
for i in range(5):
    if variant == "call":
        l = gs()
    else:
        s_ = make_samples(); gs.x.copy_(s_.tensors); gs.mask.copy_(s_.mask)
        gs.num_boxes.copy_(crit.global_num_boxes(targets, dev))
        gs.graph.replay()
        crit.iter_device += 4
        torch.cuda.synchronize()
        if variant == "readgrads":
            badg = sum(1 for p in params if not torch.isfinite(p.grad).all())
        torch.nn.utils.clip_grad_norm_(params, 0.1, foreach=True)
        opt.step()
        l = gs.loss
    print(i, variant, float(l))
print("params finite:", all(bool(torch.isfinite(p).all()) for p in model.parameters()))
bufs_ok = all(bool(torch.isfinite(b).all()) for b in model.buffers() if b.dtype.is_floating_point)
print("buffers finite:", bufs_ok)
from ocpg_amd.util.misc import NestedTensor
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    crit.iter_device = None
    o = model(make_samples(), text, targets)
    ld, *_ = crit(o, targets)
print("eager fwd after graph steps:", {k: float(v) for k, v in list(ld.items())[:6]})
st = gs.static
print("static out NaN:", [k for k, v in st["out"].items() if isinstance(v, torch.Tensor) and v.dtype.is_floating_point and not torch.isfinite(v).all()])
print("static loss NaN:", [k for k, v in st["loss_dict"].items() if not torch.isfinite(v).all()])
