"""Shared body of the product-vs-golden model checks (run on CPU with the MSDA test double, on GPU with the HIP op)."""
import torch

import cases
import synth


MASK_LOGIT_ATOL = 1e-3   # BASELINE.json north_star: "mask logits within 1e-3 fp32"


def build_product(meta, device, **over):
    from ocpg_amd.models import build_model
    cfg = dict(meta["cfg"])
    cfg.update(over)
    args = cases.default_args(device=str(device), **cfg)
    model, crit, _ = build_model(args)
    sd = synth.synth_state_dict({k: v for k, v in meta["float_shapes"].items()}, seed=meta["seed"])
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    for m in model.modules():          # parity runs: all dropout off on both sides (see make_fixtures.build_tiny)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return args, model.to(device), crit.to(device)


def text_for(B, device):
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    f, s, m = cases.tiny_text(B)
    return PrecomputedText(f.to(device), s.to(device), m.to(device))


def to_channels_last(model):
    """bench.py's layout: every Conv2d in channels-last (the ResNet body then runs NHWC end to end: fused frozen-BN NHWC
    kernels, 1x1 convs as GEMMs)."""
    for m in model.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.to(memory_format=torch.channels_last)
    return model


def run_train_step(g, tag, device, rtol, atol, channels_last=False, tag_masks=False):
    from ocpg_amd.util.misc import NestedTensor, tag_rect_mask
    meta = g.meta
    args, model, crit = build_product(meta, device)
    if channels_last:
        to_channels_last(model)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"], device)
    if tag_masks:       # what the collate helpers do: declare every frame's valid extent (memoised mask-only tensors, and the
        #                 "no padding anywhere" shortcut of the transformer when all frames fill the map)
        tag_rect_mask(mask, [hw for hw in meta[f"{tag}_sizes"] for _ in range(T)])
    model.train(), crit.train()
    out = model(NestedTensor(x, mask), text_for(B, device), targets)
    losses, *_ = crit(out, targets)
    wd = crit.weight_dict
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    res = {}
    # integer assignment: bit-exact
    assert torch.equal(torch.cat([i[0] for i in out["main_matcher_index"]]).cpu(), g[f"{tag}_main_idx"].flatten())
    aux = torch.stack([torch.cat([i[0] for i in a]) for a in out["aux_matcher_index"]]).cpu()
    assert torch.equal(aux, g[f"{tag}_aux_idx"].flatten(1))
    for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low", "ls_features"):
        got, exp = out[k].detach().cpu().float(), g[f"{tag}_{k}"]
        if k == "ls_features":
            # channel 11 is dot(f, t) / (cos(f, t) + 1e-5) (ocpg.py:376): unbounded where the cosine crosses zero, and
            # dropped by the level-set loss (criterion.py:175).  Checked by quantile; channels 0..10 element-wise.
            rel = (got[:, :, 11] - exp[:, :, 11]).abs() / (exp[:, :, 11].abs() + 1e-3)
            assert (rel > 1e-3).float().mean().item() < 0.01
            got, exp = got[:, :, :11], exp[:, :, :11]
        err = (got - exp).abs().max().item()
        res[k] = err
        # mask logits: the north-star bound (<= 1e-3 absolute, fp32); everything else rtol/atol
        a = MASK_LOGIT_ATOL if k in ("pred_masks", "pred_masks_low") else atol
        assert torch.allclose(got, exp, rtol=rtol, atol=a), f"{tag} {k}: max abs err {err:.3e}"
    for i, a in enumerate(out["aux_outputs"]):
        assert torch.allclose(a["pred_masks"].detach().cpu().float(), g[f"{tag}_aux{i}_pred_masks"], rtol=rtol, atol=MASK_LOGIT_ATOL)
        assert torch.allclose(a["pred_boxes"].detach().cpu().float(), g[f"{tag}_aux{i}_pred_boxes"], rtol=rtol, atol=atol)
    ref_losses = meta[f"{tag}_losses"]
    assert set(losses) == set(ref_losses)
    for k, v in ref_losses.items():
        assert abs(losses[k].item() - v) <= 10 * rtol * abs(v) + 10 * atol, (k, losses[k].item(), v)
    assert abs(total.item() - g[f"{tag}_total"].item()) <= 10 * rtol * abs(g[f"{tag}_total"].item())
    params = dict(model.named_parameters())
    gn_ref = meta[f"{tag}_grad_norms"]
    bad = []
    for k, v in gn_ref.items():
        gr = params[k].grad
        if v is None:
            assert gr is None, k
            continue
        assert gr is not None, k
        n = gr.norm().item()
        if abs(n - v) > 50 * rtol * abs(v) + 50 * atol:
            bad.append((k, n, v))
    assert not bad, bad[:8]
    for key in g.keys():
        if key.startswith(f"{tag}_grad_"):
            name = key[len(f"{tag}_grad_"):]
            got, exp = params[name].grad.detach().cpu(), g[key]
            # L2-relative: a sampling point that sits within an ulp of a pixel boundary flips its bilinear cell under
            # any re-association upstream and moves single entries of the offset gradients by a few percent
            rel = (got - exp).norm().item() / (exp.norm().item() + 1e-12)
            assert rel <= 5e-2, (name, rel)
    res["total"] = total.item()
    return res


def run_eval(g, tag, device, rtol, atol):
    from ocpg_amd.util.misc import NestedTensor
    meta = g.meta
    args, model, crit = build_product(meta, device)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"], device)
    model.eval()
    with torch.no_grad():
        out = model(NestedTensor(x, mask), text_for(B, device), targets)
    for k in ("pred_logits", "pred_boxes", "pred_masks", "reference_points"):
        got, exp = out[k].cpu().float(), g[f"{tag}_eval_{k}"]
        assert got.shape == exp.shape, (k, got.shape, exp.shape)
        a = MASK_LOGIT_ATOL if k == "pred_masks" else atol
        assert torch.allclose(got, exp, rtol=rtol, atol=a), f"{tag} eval {k}: {(got - exp).abs().max().item():.3e}"


def reference_optimizer(model, args):
    """main.py:76-99: AdamW, four name-based LR groups."""
    def has(n, keys):
        return any(k in n for k in keys)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [
        {"params": [p for n, p in named if not has(n, args.lr_backbone_names) and not has(n, args.lr_text_encoder_names)
                    and not has(n, args.lr_linear_proj_names)], "lr": args.lr},
        {"params": [p for n, p in named if has(n, args.lr_backbone_names)], "lr": args.lr_backbone},
        {"params": [p for n, p in named if has(n, args.lr_text_encoder_names)], "lr": args.lr_text_encoder},
        {"params": [p for n, p in named if has(n, args.lr_linear_proj_names)], "lr": args.lr * args.lr_linear_proj_mult},
    ]
    return torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay), named


def check_reference_iteration(g, device):
    """tests/golden/train_step.npz = ONE iteration of the reference's engine.train_one_epoch (make_fixtures.gen_train_step)."""
    from ocpg_amd import engine
    from ocpg_amd.util.misc import NestedTensor
    meta = g.meta
    args, model, crit = build_product(meta, device)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta["sizes"], device)
    model.train(), crit.train()
    opt, named = reference_optimizer(model, args)
    assert [g_["lr"] for g_ in opt.param_groups] == meta["group_lrs"]
    assert [len(g_["params"]) for g_ in opt.param_groups] == meta["group_sizes"]          # same tensors in the same LR groups
    first = [next((n for n, p in named if p is g_["params"][0]), None) if g_["params"] else None for g_ in opt.param_groups]
    assert first == meta["group_first_names"]
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    loss, _, norm = engine.train_step(model, crit, NestedTensor(x, mask), text_for(B, device), targets, opt, max_norm=meta["clip_max_norm"])
    assert abs(loss - meta["loss"]) <= 2e-3 * abs(meta["loss"]), (loss, meta["loss"])
    assert abs(float(norm) - meta["grad_norm"]) <= 2e-2 * meta["grad_norm"], (float(norm), meta["grad_norm"])
    after = dict(model.named_parameters())
    # post-step parameters: |p|_1 of every tensor, the update norm of every tensor, and whole tensors for a few of them
    bad = []
    for k, v in meta["param_abs_sum"].items():
        got = float(after[k].detach().double().abs().sum())
        if abs(got - v) > 1e-4 * abs(v) + 1e-6:
            bad.append((k, got, v))
    assert not bad, bad[:5]
    bad = []
    for k, v in meta["param_delta_norm"].items():
        got = float((after[k].detach() - before[k]).double().norm())
        # AdamW's first step moves every element by ~lr * sign(g): the update norm is insensitive to |g|, so this pins
        # the group's LR, the weight decay and which tensors were stepped; elements with |g| ~ eps may differ
        if abs(got - v) > 2e-2 * abs(v) + 1e-7:
            bad.append((k, got, v))
    assert len(bad) <= max(2, len(meta["param_delta_norm"]) // 100), bad[:8]
    for key in g.keys():
        if key.startswith("after_"):
            name = key[len("after_"):]
            # whole tensors: AdamW's first step moves an element by ~lr * g / (|g| + eps), so an element whose gradient is at
            # rounding level may land anywhere within +-lr of the reference's value; everything else must agree
            d = (after[name].detach().cpu() - g[key]).abs()
            off = d > (1e-4 * g[key].abs() + 2e-5)
            lr_max = max(meta["group_lrs"])
            assert off.float().mean().item() <= 0.01 and d.max().item() <= 2.1 * lr_max, (name, int(off.sum()), d.numel(), d.max().item())


def check_reference_checkpoint(g, device):
    """tests/golden/ckpt_ref.pth was WRITTEN BY THE REFERENCE (util.misc.save_on_master with main.py:229-236's dict, head modules
    only to keep it committable).  It must load through util.checkpoint into the product's model / optimizer / scheduler, resume
    at the decayed LR, and a file the product writes must carry the same container format and keys."""
    import os
    from ocpg_amd.util import checkpoint as ck
    meta = g.meta
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_ref.pth")
    with open(path, "rb") as f:
        assert f.read(2) != b"PK"                                  # legacy (non-zip) container
    state = torch.load(path, map_location="cpu", weights_only=False)
    assert set(state) == {"model", "optimizer", "lr_scheduler", "epoch", "args", "grad_scaler"}
    assert list(state["model"]) == meta["keys"] and state["epoch"] == meta["epoch"]
    args = cases.default_args(device=str(device), **cases.TINY)
    from ocpg_amd.models import build_model
    model, _, _ = build_model(args)
    model.to(device)
    named = dict(model.named_parameters())
    assert all(k in named and list(named[k].shape) == meta["shapes"][k] for k in meta["keys"])
    sub = [(k, named[k]) for k in meta["keys"]]
    opt = torch.optim.AdamW([{"params": [p for n, p in sub if "reference_points" not in n], "lr": args.lr},
                             {"params": [p for n, p in sub if "reference_points" in n], "lr": args.lr * args.lr_linear_proj_mult}],
                            lr=args.lr, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [1, 5])
    missing, unexpected, epoch = ck.load_checkpoint(state, model, opt, sched)
    assert not unexpected and epoch == meta["epoch"] and set(missing).isdisjoint(meta["keys"])
    for k in meta["keys"]:
        assert abs(float(named[k].detach().double().abs().sum()) - meta["abs_sum"][k]) <= 1e-6 * abs(meta["abs_sum"][k]) + 1e-9, k
    got = [float(opt.state[p]["exp_avg"].double().abs().sum()) for _, p in sub]
    assert all(abs(a - b) <= 1e-6 * abs(b) + 1e-12 for a, b in zip(got, meta["exp_avg_abs_sum"]))
    assert sched.last_epoch == meta["last_epoch"]
    assert all(abs(a["lr"] - b) <= 1e-15 for a, b in zip(opt.param_groups, meta["lrs_after_resume"])), ([g_["lr"] for g_ in opt.param_groups], meta["lrs_after_resume"])


def check_collate(g):
    """util/misc.py:299-379 on ragged clips, against tensors produced by the reference's own functions (infer_collate.npz)."""
    from ocpg_amd.util import misc
    m = g.meta
    clips = [synth.rand(f"col_clip{i}", tuple(s)) for i, s in enumerate(m["collate_sizes"])]
    samples, tg = misc.collate_fn(list(zip(clips, [{"k": i} for i in range(len(clips))])))
    assert [t["k"] for t in tg] == [0, 1, 2] and isinstance(tg, tuple)
    assert torch.equal(samples.tensors, g["collate_tensors"]) and torch.equal(samples.mask, g["collate_mask"])
    assert misc.mask_key(samples.mask) is not None          # the collate step declares every frame's valid extent
    flat = [synth.rand(f"col_img{i}", tuple(s)) for i, s in enumerate(m["split_sizes"])]
    nt = misc.nested_tensor_from_tensor_list(flat, size_divisibility=8, split=True)
    assert torch.equal(nt.tensors, g["split_tensors"]) and torch.equal(nt.mask, g["split_mask"])
    nt1 = misc.nested_tensor_from_videos_list(clips[:2], size_divisibility=1)
    assert torch.equal(nt1.tensors, g["nodiv_tensors"]) and torch.equal(nt1.mask, g["nodiv_mask"])


def check_inference_loop(g, device, atol_logits=1e-4, atol_masks=1e-4, label_mismatch=0.0):
    """inference.segment_video + merge_objects against the outputs of the reference's own inference_davis.py:203-261 statements
    (two expressions of one 5-frame video, clips of 2 + 2 + 1 frames, 160 x 200 frames -> padded 160 x 224 -> 320 x 400 masks)."""
    from ocpg_amd import inference
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    m = g.meta
    args, model, _ = build_product(m, device)
    frames = g["infer_frames"].to(device)
    f, s, pm = cases.tiny_text(1)
    masks = []
    for obj, (a, b) in enumerate(m["text_scale"]):
        text = PrecomputedText((f * a).to(device), (s * b).to(device), pm.to(device))
        logits, mk = inference.segment_video(model, frames, text, clip_len=m["crop_len"], origin_size=tuple(m["origin"]))
        assert logits.shape == g[f"infer_logits{obj}"].shape and mk.shape == g[f"infer_masks{obj}"].shape
        el = (logits.float().cpu() - g[f"infer_logits{obj}"]).abs().max().item()
        em = (mk.float().cpu() - g[f"infer_masks{obj}"]).abs().max().item()
        assert el <= atol_logits and em <= atol_masks, (obj, el, em)
        masks.append(mk.float().cpu())
    lab = inference.merge_objects(torch.stack(masks))
    bad = (lab != g["infer_labels"]).float().mean().item()
    assert bad <= label_mismatch, bad
