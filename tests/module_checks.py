"""Per-module product-vs-golden checks (CPU with the MSDA test double; GPU with the HIP op)."""
import torch

import cases
import synth


def close(got, exp, rtol, atol, what):
    got, exp = got.detach().cpu().to(exp.dtype), exp
    err = (got - exp).abs().max().item()
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    assert torch.allclose(got, exp, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e} (ref max {exp.abs().max().item():.3e})"


def check_lfm(g, dev, rtol=1e-4, atol=1e-5):
    from ocpg_amd.models.modules import LFMResizeAdaptive
    m = g.meta
    mod = LFMResizeAdaptive(m["c"], m["sigma"])
    mod.load_state_dict(synth.synth_state_dict(m["param_shapes"], seed=m["seed"]))
    mod.to(dev)
    xs = [synth.rand(f"lfm_x{i}", s).to(dev).requires_grad_(True) for i, s in enumerate([(3, m["c"], 12, 20), (3, m["c"], 6, 10), (3, m["c"], 3, 5)])]
    y0, g0 = mod(xs[0])
    y1, g1 = mod(xs[1], g0)
    y2, g2 = mod(xs[2], g1)
    for k, v in (("y0", y0), ("g0", g0), ("y1", y1), ("g1", g1), ("y2", y2), ("g2", g2)):
        close(v, g[k], rtol, atol, "lfm " + k)
    loss = sum((y * synth.rand(f"lfm_go{i}", y.shape).to(dev)).sum() for i, y in enumerate((y0, y1, y2)))
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(loss, xs + list(params.values()))
    for i in range(3):
        close(grads[i], g[f"gx{i}"], rtol * 10, atol * 10, f"lfm gx{i}")
    for (k, _), gr in zip(params.items(), grads[3:]):
        close(gr, g["gp_" + k], rtol * 10, atol * 50, "lfm gp_" + k)


def check_fusion(g, dev, rtol=1e-4, atol=1e-5):
    from ocpg_amd.models.segmentation import VisionLanguageFusionModule
    m = g.meta
    mod = VisionLanguageFusionModule(d_model=m["d"], nhead=m.get("nhead", 8))
    mod.load_state_dict(synth.synth_state_dict(m["param_shapes"], seed=m["seed"]))
    mod.to(dev)
    t, h, w, b, Lt, d = m["t"], m["h"], m["w"], m["b"], m["Lt"], m["d"]
    vis = synth.rand("fu_vis", (t, h, w, b, d)).to(dev).requires_grad_(True)
    text = synth.rand("fu_text", (Lt, b, d)).to(dev).requires_grad_(True)
    tpos = synth.rand("fu_tpos", (Lt, b, d)).to(dev)
    out = mod(visual=vis, text=text, text_key_padding_mask=g["pad"].to(dev), text_pos=tpos, visual_pos=None)
    close(out, g["out"], rtol, atol, "fusion out")
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad((out * synth.rand("fu_go", out.shape).to(dev)).sum(), [vis, text] + list(params.values()))
    close(grads[0], g["gvis"], rtol * 10, atol * 10, "fusion gvis")
    close(grads[1], g["gtext"], rtol * 10, atol * 10, "fusion gtext")
    for (k, _), gr in zip(params.items(), grads[2:]):
        close(gr, g["gp_" + k], rtol * 10, atol * 50, "fusion gp_" + k)


def check_msda_module(g, dev, rtol=1e-4, atol=1e-5, host_shapes=False):
    """host_shapes: the level shapes also travel as a host copy on the tensor (what the product's own transformer does): the self-attention
    case (r2: Lq == S) then takes the fused front end + the self-attention backward kernels instead of the generic entry points."""
    from ocpg_amd.models.ops.modules import MSDeformAttn
    m = g.meta
    d, L, M, P, N, Lq = m["d"], m["L"], m["M"], m["P"], m["N"], m["Lq"]
    shapes, lsi = cases.level_start([tuple(s) for s in m["shapes"]])
    S = int(shapes.prod(1).sum())
    mod = MSDeformAttn(d, L, M, P)
    mod.load_state_dict(synth.synth_state_dict(m["param_shapes"], seed=m["seed"]))
    mod.to(dev)
    pad = g["pad"].to(dev)
    src = synth.rand("mm_src", (N, S, d)).to(dev).requires_grad_(True)
    shapes_d, lsi_d = shapes.to(dev), lsi.to(dev)
    if host_shapes:
        shapes_d._ocpg_host = shapes
    for tag, refdim, q_len in (("r2", 2, S), ("r4", 4, Lq)):
        query = synth.rand("mm_q" + tag, (N, q_len, d)).to(dev).requires_grad_(True)
        ref = synth.rand("mm_ref" + tag, (N, q_len, L, refdim), uniform=True)
        if refdim == 4:
            ref = torch.cat([ref[..., :2], ref[..., 2:] * 0.4 + 0.05], -1)
        out, loc, attn = mod(query, ref.to(dev), src, shapes_d, lsi_d, pad)
        close(out, g[f"{tag}_out"], rtol, atol, tag + " out")
        close(loc, g[f"{tag}_loc"], rtol, atol, tag + " loc")
        close(attn, g[f"{tag}_attn"], rtol, atol, tag + " attn")
        params = list(mod.named_parameters())
        grads = torch.autograd.grad((out * synth.rand("mm_go" + tag, out.shape).to(dev)).sum(), [query, src] + [p for _, p in params])
        close(grads[0], g[f"{tag}_gq"], rtol * 10, atol * 10, tag + " gq")
        close(grads[1], g[f"{tag}_gsrc"], rtol * 10, atol * 10, tag + " gsrc")
        for (k, _), gr in zip(params, grads[2:]):
            close(gr, g[f"{tag}_gp_{k}"], rtol * 10, atol * 100, f"{tag} gp_{k}")


def check_transformer(g, dev, rtol=2e-4, atol=2e-5):
    from ocpg_amd.models.deformable_transformer import DeformableTransformer
    from ocpg_amd.models.ocpg import MLP, _get_clones
    m = g.meta
    d, L = m["d"], m["L"]
    tr = DeformableTransformer(d_model=d, nhead=m.get("nhead", 8), num_encoder_layers=m["enc"], num_decoder_layers=m["dec"], dim_feedforward=m["ffn"],
                               dropout=0.0, return_intermediate_dec=True, num_feature_levels=L, dec_n_points=4, enc_n_points=4)
    tr.decoder.bbox_embed = _get_clones(MLP(d, d, 4, 3), m["dec"])
    tr.load_state_dict(synth.synth_state_dict(m["param_shapes"], seed=m["seed"]))
    tr.to(dev)
    B, T, Q = m["B"], m["T"], m["Q"]
    N = B * T
    shapes_l = [tuple(s) for s in m["shapes"]]
    masks = [g[f"mask{i}"].to(dev) for i in range(L)]
    srcs = [synth.rand(f"tr_src{i}", (N, d, h, w)).to(dev).requires_grad_(True) for i, (h, w) in enumerate(shapes_l)]
    poss = [synth.rand(f"tr_pos{i}", (N, d, h, w)).to(dev) for i, (h, w) in enumerate(shapes_l)]
    tgt = synth.rand("tr_tgt", (B, T, Q, d)).to(dev)
    qe = synth.rand("tr_qe", (Q, d)).to(dev)
    hs, memory, init_ref, inter_ref, _, _, inter_samples = tr(srcs, tgt, masks, poss, qe)
    close(hs, g["hs"], rtol, atol, "hs")
    close(init_ref, g["init_ref"], rtol, atol, "init_ref")
    close(inter_ref, g["inter_ref"], rtol, atol, "inter_ref")
    close(inter_samples, g["inter_samples"], rtol, atol * 10, "inter_samples")
    for i, mem in enumerate(memory):
        close(mem, g[f"memory{i}"], rtol, atol, f"memory{i}")
    loss = (hs * synth.rand("tr_go", hs.shape).to(dev)).sum() + sum((mm * synth.rand(f"tr_gm{i}", mm.shape).to(dev)).sum() for i, mm in enumerate(memory))
    params = dict(tr.named_parameters())
    grads = torch.autograd.grad(loss, srcs + list(params.values()), allow_unused=True)
    for i in range(L):
        close(grads[i], g[f"gsrc{i}"], rtol * 10, atol * 20, f"gsrc{i}")
    gn = m["grad_norms"]
    for (k, _), gr in zip(params.items(), grads[L:]):
        if gn[k] is None:
            assert gr is None or gr.abs().max() == 0, k
        else:
            n = gr.norm().item()
            assert abs(n - gn[k]) <= 1e-3 * abs(gn[k]) + 1e-5, (k, n, gn[k])


def check_dynmask_mso(g, dev, rtol=1e-4, atol=1e-4):
    from ocpg_amd.models import build_model
    from ocpg_amd.util.misc import NestedTensor
    m = g.meta
    args = cases.default_args(device=str(dev), **m["cfg"])
    model, _, _ = build_model(args)
    b, t, q, c, h, w = m["b"], m["t"], m["q"], m["c"], m["h"], m["w"]
    feats = synth.rand("dm_feat", (b, t, c, h, w)).to(dev).requires_grad_(True)
    params = (synth.rand("dm_par", (b, t * q, m["n_par"])) * 0.2).to(dev).requires_grad_(True)
    refp = synth.rand("dm_ref", (b, t * q, 2), uniform=True).to(dev).requires_grad_(True)
    targets = [{"size": torch.tensor(s).to(dev)} for s in m["sizes"]]
    model.to(dev)
    out = model.dynamic_mask_with_coords(feats, params, refp, targets)
    close(out, g["dm_out"], rtol, atol, "dynmask out")
    gf, gp, gr = torch.autograd.grad((out * synth.rand("dm_go", out.shape).to(dev)).sum(), (feats, params, refp))
    close(gf, g["dm_gfeat"], rtol * 10, atol * 10, "dynmask gfeat")
    close(gp, g["dm_gpar"], rtol * 10, atol * 100, "dynmask gpar")
    close(gr, g["dm_gref"], rtol * 10, atol * 100, "dynmask gref")
    mso = model.mask_refine
    mso.load_state_dict(synth.synth_state_dict(m["mso_param_shapes"], seed=m["seed"]))
    # the fixture's MSO weights were those of the whole tiny model (key prefix 'mask_refine.')
    mso.load_state_dict({k: synth.synth_tensor("mask_refine." + k, s, m["seed"]) for k, s in m["mso_param_shapes"].items()})
    mso.to(dev)
    n = 3
    pm = synth.rand("mso_pm", (n, 16, 6, 7)).to(dev).requires_grad_(True)
    f4 = synth.rand("mso_f4", (n, 256, 12, 14)).to(dev).requires_grad_(True)
    f8 = synth.rand("mso_f8", (n, 512, 6, 7)).to(dev).requires_grad_(True)
    feats_l = [NestedTensor(f4, None), NestedTensor(f8, None)]
    o = mso(pm, feats_l)
    close(o, g["mso_out"], rtol, atol, "mso out")
    o2 = mso.forward_multi([pm, pm * 0.5], feats_l)
    close(o2[0], g["mso_out"], rtol, atol, "mso multi out")
    mp = dict(mso.named_parameters())
    grads = torch.autograd.grad((o2[0] * synth.rand("mso_go", o.shape).to(dev)).sum(), [pm, f4, f8] + list(mp.values()))
    close(grads[0], g["mso_gpm"], rtol * 10, atol * 10, "mso gpm")
    close(grads[1], g["mso_gf4"], rtol * 10, atol * 10, "mso gf4")
    close(grads[2], g["mso_gf8"], rtol * 10, atol * 10, "mso gf8")
    for (k, _), gr_ in zip(mp.items(), grads[3:]):
        close(gr_, g["mso_gp_" + k], rtol * 10, atol * 100, "mso gp_" + k)


def check_matcher_crit(g, dev, rtol=1e-4, atol=1e-5):
    from ocpg_amd.models import build_model
    m = g.meta
    args = cases.default_args(device=str(dev), **m["cfg"])
    model, crit, _ = build_model(args)
    crit.to(dev)
    b, t, q, H, W = m["b"], m["t"], m["q"], m["H"], m["W"]
    targets = synth.synthetic_targets(b, t, H, W)
    targets[1]["valid"] = torch.tensor([1, 0])
    targets[1]["boxes"] = torch.tensor([[0.3, 0.4, 0.2, 0.3], [0.6, 0.5, 0.3, 0.2]])
    targets[0]["weights"] = synth.rand("mc_heat", (t, H, W), uniform=True) * targets[0]["masks"]
    targets = [{k: v.to(dev) for k, v in tg.items()} for tg in targets]
    outputs = {
        "pred_logits": synth.rand("mc_logits", (b, t, q, 1)).to(dev).requires_grad_(True),
        "pred_boxes": (synth.rand("mc_boxes", (b, t, q, 4), uniform=True) * 0.5 + 0.2).to(dev).requires_grad_(True),
        "pred_masks": synth.rand("mc_masks", (b, t, q, H // 2, W // 2)).to(dev),
    }
    ind = model.matcher(outputs, targets)
    assert torch.equal(torch.stack([i[0] for i in ind]).cpu(), g["idx_src"])      # integer result: bit-exact
    assert torch.equal(torch.stack([i[1] for i in ind]).cpu(), g["idx_tgt"])
    out = {
        "pred_logits": outputs["pred_logits"], "pred_boxes": outputs["pred_boxes"],
        "pred_masks": synth.rand("mc_pm", (b, t, H, W)).to(dev).requires_grad_(True),
        "pred_masks_low": synth.rand("mc_pml", (b, t, H // 2, W // 2)).to(dev).requires_grad_(True),
        "ls_features": synth.rand("mc_ls", (b, t, 12, H // 2, W // 2)).to(dev).requires_grad_(True),
        "frames": synth.rand("mc_fr", (b, t, 3, H // 2, W // 2)).to(dev),
        "main_matcher_index": ind, "aux_matcher_index": [],
    }
    losses, src_m, tgt_m, weak_m = crit(out, targets)
    wd = crit.weight_dict
    assert {k: float(v) for k, v in wd.items()} == m["weight_dict"]
    for k, v in losses.items():
        close(v, g["loss_" + k], rtol, atol, "loss " + k)
    assert set("loss_" + k for k in losses) == {k for k in g.keys() if k.startswith("loss_")}
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    close(total, g["total"], rtol, atol, "total")
    # the one-reduction total the bench loop uses, on the criterion's own dict and (generic path) on a copy of it
    close(crit.weighted_sum(losses), g["total"], rtol, atol, "weighted_sum")
    close(crit.weighted_sum(dict(losses)), g["total"], rtol, atol, "weighted_sum (foreign dict)")
    total = crit.weighted_sum(losses)
    leaves = [out["pred_logits"], out["pred_boxes"], out["pred_masks"], out["pred_masks_low"], out["ls_features"]]
    grads = torch.autograd.grad(total, leaves)
    for gr, k in zip(grads, ("g_logits", "g_boxes", "g_pm", "g_pml", "g_ls")):
        close(gr, g[k], rtol * 10, atol, k)
    close(src_m, g["src_m"], rtol, atol, "src_m")
    close(tgt_m, g["tgt_m"], rtol, atol, "tgt_m")
    close(weak_m, g["weak_m"], rtol, atol, "weak_m")
