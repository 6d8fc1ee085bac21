"""Diagnostic: forward time per model section (cuda events via module hooks) + total fwd / criterion / bwd / optimizer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from ocpg_amd.models import build_model

dev = torch.device("cuda:0")
args = bench.model_args(dev, "resnet101", amp=True)
model, crit, _ = build_model(args)
model.to(dev).to(memory_format=torch.channels_last); crit.to(dev); model.train(); crit.train()
opt = bench.make_optimizer(model, args)
make_samples, text, targets = bench.synthetic_batch(2, dev, 42)
sections = {"backbone": model.backbone, "input_proj": model.input_proj, "input_fft": model.input_fft, "input_fft_post": model.input_fft_post,
            "fusion": model.fusion_module, "encoder": model.transformer.encoder, "decoder": model.transformer.decoder,
            "controller": model.controller, "mask_refine": model.mask_refine, "matcher": model.matcher, "criterion": crit,
            "text_proj": model.text_proj, "bbox_embed": model.bbox_embed, "class_embed": model.class_embed}
ev = []
def pre(name):
    def f(m, i):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, 0, e))
    return f
def post(name):
    def f(m, i, o):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, 1, e))
    return f
for n, m in sections.items():
    mods = list(m) if isinstance(m, torch.nn.ModuleList) else [m]
    for mm in mods:
        mm.register_forward_pre_hook(pre(n)); mm.register_forward_hook(post(n))

def step(timed):
    global ev
    ev = []
    E = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    E[0].record()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(make_samples(), text, targets)
        E[1].record()
        ld, *_ = crit(out, targets)
        loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    E[2].record()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    E[3].record()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1, foreach=True)
    opt.step()
    E[4].record()
    torch.cuda.synchronize()
    if timed:
        print("fwd %.1f  criterion %.1f  bwd %.1f  clip+opt %.1f  total %.1f ms" % (E[0].elapsed_time(E[1]), E[1].elapsed_time(E[2]), E[2].elapsed_time(E[3]), E[3].elapsed_time(E[4]), E[0].elapsed_time(E[4])))
        acc = {}
        stack = {}
        for name, kind, e in ev:
            if kind == 0:
                stack.setdefault(name, []).append(e)
            else:
                s = stack[name].pop()
                acc[name] = acc.get(name, 0.0) + s.elapsed_time(e)
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
            print("  %-16s %7.2f ms" % (k, v))
for i in range(4):
    step(False)
t0 = time.perf_counter(); step(True); step(True)
