#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE (TJUMMG/OCPG) on CPU.

Run in the build container only:  ``python tests/golden/make_fixtures.py [name ...]``
The reference is imported through ``ref_import.py`` (stand-ins documented there); only inputs that
cannot be regenerated from a seed, and the reference's outputs, are written (``*.npz``).  No reference
source text is stored.  Fixture <-> reference map:

  msda_testpy      models/ops/test.py:21-60 protocol (seed 3, shapes, fp64 + fp32) and its gradient cases
  msda_cases       ms_deform_attn_core_pytorch (ms_deform_attn_func.py:41-61) on larger seeded cases, incl.
                   sampling points outside the map (zero padding) -- out + grads (fp64 truth, fp32)
  msda_module      MSDeformAttn.forward (ms_deform_attn.py:80-118), 2-d and 4-d reference points, padding mask
  transformer      DeformableTransformer.forward (deformable_transformer.py:134-217), 2 enc + 2 dec layers
  lfm              LFMResizeAdaptive.forward (modules.py:33-61) without / with incoming gauss_map
  fusion           VisionLanguageFusionModule.forward (segmentation.py:103-113) with key padding
  *_d32            msda_module / transformer / fusion again with 2 heads of head_dim 32: the shapes the production HIP kernels serve
  dynmask_mso      OCPG.dynamic_mask_with_coords (ocpg.py:475-529) and MSO.forward (decoder.py:31-46)
  matcher_crit     HungarianMatcher.forward (matcher.py:74-171) and SetCriterion.forward (criterion.py:213-254)
  e2e_tiny         OCPG.forward + criterion + backward (ocpg.py:197-447), train mode, with / without padding,
                   plus the eval-mode tail (ocpg.py:401-433)
  e2e_d32          the same step with head_dim 32 (the MSDeformAttn kernels of the BASELINE configurations)
  e2e_cfg1         BASELINE config #1 shapes: one frame, 256x256, 3 feature levels, 1 query
  train_step       one iteration of engine.train_one_epoch (engine.py:29-123) with main.py:76-99's optimizer: loss, gradient
                   norm, post-step parameters; ckpt_ref.pth = a checkpoint written by util.misc.save_on_master (main.py:229-236)
  swin_n392        WindowAttention3D at the full (8,7,7) window, N = 392 (config #5), head_dim 32, with shift mask
  swin3d           Video-Swin pieces (video_swin_transformer.py) -- see gen_swin3d
  infer_collate    util.misc.collate_fn & friends (util/misc.py:299-379) and the per-video core of inference_davis.py:203-261, whose
                   statements are read from the reference checkout and executed at generation time
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import ref_import  # noqa: E402
import synth  # noqa: E402
from cases import MSDA_CASES, TINY, level_start, msda_case_inputs, padded_masks, tiny_text  # noqa: E402

torch.set_num_threads(8)


def save(name, meta, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    out["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays)")


# ----------------------------------------------------------------------------------------------
def gen_msda_testpy():
    ref_import.install()
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as core
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes, lsi = level_start([(6, 4), (3, 2)])
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    arrays = {"shapes": shapes, "level_start": lsi}

    def draw(ch):
        value = torch.rand(N, S, M, ch) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        return value, loc, attn

    v, l, a = draw(D)                                    # check_forward_equal_with_pytorch_double
    arrays.update(d_value=v, d_loc=l, d_attn=a, d_out=core(v.double(), shapes, l.double(), a.double()))
    v, l, a = draw(D)                                    # check_forward_equal_with_pytorch_float
    arrays.update(f_value=v, f_loc=l, f_attn=a, f_out=core(v, shapes, l, a))
    chans = [30, 32, 64, 71, 1025]                       # check_gradient_numerical channel list (2048/3096 in the GPU test only)
    for ch in chans:
        v, l, a = draw(ch)
        vd, ld, ad = (x.double().requires_grad_(True) for x in (v, l, a))
        out = core(vd, shapes, ld, ad)
        g = synth.rand(f"testpy_go_{ch}", out.shape).double()
        gv, gl, ga = torch.autograd.grad((out * g).sum(), (vd, ld, ad))
        arrays.update({f"g{ch}_value": v, f"g{ch}_loc": l, f"g{ch}_attn": a, f"g{ch}_out": out,
                       f"g{ch}_gv": gv, f"g{ch}_gl": gl, f"g{ch}_ga": ga})
    save("msda_testpy", {"N": N, "M": M, "D": D, "Lq": Lq, "L": L, "P": P, "grad_channels": chans,
                         "grad_out": "synth.rand('testpy_go_<ch>', out.shape).double()"}, **arrays)


def gen_msda_cases():
    ref_import.install()
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as core
    arrays = {}
    for c in MSDA_CASES:
        value, shapes, lsi, loc, attn, go = msda_case_inputs(c)
        for tag, dt in (("64", torch.float64), ("32", torch.float32)):
            v, l, a = (x.to(dt).requires_grad_(True) for x in (value, loc, attn))
            out = core(v, shapes, l, a)
            gv, gl, ga = torch.autograd.grad((out * go.to(dt)).sum(), (v, l, a))
            n = c["name"]
            if tag == "64":
                arrays.update({f"{n}_out64": out, f"{n}_gv64": gv, f"{n}_gl64": gl, f"{n}_ga64": ga})
            else:
                arrays.update({f"{n}_out32": out})
    save("msda_cases", {"cases": MSDA_CASES}, **arrays)


# ----------------------------------------------------------------------------------------------
def gen_msda_module(name="msda_module", M=8):
    ref_import.install()
    from models.ops.modules import MSDeformAttn
    d, L, P = 64, 4, 4
    shapes_l = [(8, 12), (4, 6), (2, 3), (1, 2)]
    shapes, lsi = level_start(shapes_l)
    S = int(shapes.prod(1).sum())
    N, Lq = 2, 9
    mod = MSDeformAttn(d, L, M, P)
    sd = synth.synth_state_dict(synth.shapes_of(mod), seed=2)
    mod.load_state_dict(sd)
    arrays = {}
    pad = torch.cat([m.flatten(1) for m in padded_masks(N, shapes_l, [1.0, 0.75], [1.0, 0.6])], 1)
    src = synth.rand("mm_src", (N, S, d)).requires_grad_(True)
    for tag, refdim, q_len in (("r2", 2, S), ("r4", 4, Lq)):
        query = synth.rand("mm_q" + tag, (N, q_len, d)).requires_grad_(True)
        ref = synth.rand("mm_ref" + tag, (N, q_len, L, refdim), uniform=True)
        if refdim == 4:
            ref = torch.cat([ref[..., :2], ref[..., 2:] * 0.4 + 0.05], -1)
        out, loc, attn = mod(query, ref, src, shapes, lsi, pad)
        go = synth.rand("mm_go" + tag, out.shape)
        params = list(mod.parameters())
        grads = torch.autograd.grad((out * go).sum(), [query, src] + params)
        arrays.update({f"{tag}_out": out, f"{tag}_loc": loc, f"{tag}_attn": attn, f"{tag}_gq": grads[0], f"{tag}_gsrc": grads[1]})
        for (k, _), g in zip(mod.named_parameters(), grads[2:]):
            arrays[f"{tag}_gp_{k}"] = g
    save(name, {"d": d, "L": L, "M": M, "P": P, "shapes": shapes_l, "N": N, "Lq": Lq, "seed": 2,
                         "pad_frac_h": [1.0, 0.75], "pad_frac_w": [1.0, 0.6], "param_shapes": synth.shapes_of(mod)}, pad=pad, **arrays)


def gen_transformer(name="transformer", nhead=8):
    ref_import.install()
    from models.deformable_transformer import DeformableTransformer
    d, ffn, L = 64, 128, 4
    shapes_l = [(8, 12), (4, 6), (2, 3), (1, 2)]
    tr = DeformableTransformer(d_model=d, nhead=nhead, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=ffn,
                               dropout=0.0, return_intermediate_dec=True, num_feature_levels=L,
                               dec_n_points=4, enc_n_points=4)
    from models.ocpg import MLP, _get_clones
    bbox = _get_clones(MLP(d, d, 4, 3), 2)
    tr.decoder.bbox_embed = bbox
    shp = synth.shapes_of(tr)
    tr.load_state_dict(synth.synth_state_dict(shp, seed=3))
    B, T, Q = 1, 2, 3
    N = B * T
    masks = padded_masks(N, shapes_l, [1.0, 0.75], [0.8, 1.0])
    srcs = [synth.rand(f"tr_src{i}", (N, d, h, w)).requires_grad_(True) for i, (h, w) in enumerate(shapes_l)]
    poss = [synth.rand(f"tr_pos{i}", (N, d, h, w)) for i, (h, w) in enumerate(shapes_l)]
    tgt = synth.rand("tr_tgt", (B, T, Q, d))
    qe = synth.rand("tr_qe", (Q, d))
    hs, memory, init_ref, inter_ref, _, _, inter_samples = tr(srcs, tgt, masks, poss, qe)
    go = synth.rand("tr_go", hs.shape)
    gm = [synth.rand(f"tr_gm{i}", m.shape) for i, m in enumerate(memory)]
    loss = (hs * go).sum() + sum((m * g).sum() for m, g in zip(memory, gm))
    params = dict(tr.named_parameters())
    grads = torch.autograd.grad(loss, srcs + list(params.values()), allow_unused=True)
    arrays = dict(hs=hs, init_ref=init_ref, inter_ref=inter_ref, inter_samples=inter_samples)
    for i, m in enumerate(memory):
        arrays[f"memory{i}"] = m
    for i in range(len(srcs)):
        arrays[f"gsrc{i}"] = grads[i]
    gn = {}
    for (k, _), g in zip(params.items(), grads[len(srcs):]):
        gn[k] = float(g.norm()) if g is not None else None
    for i, m in enumerate(masks):
        arrays[f"mask{i}"] = m
    save(name, {"d": d, "nhead": nhead, "ffn": ffn, "L": L, "shapes": shapes_l, "B": B, "T": T, "Q": Q, "seed": 3,
                         "enc": 2, "dec": 2, "param_shapes": shp, "grad_norms": gn}, **arrays)


def gen_lfm():
    ref_import.install()
    from models.modules import LFMResizeAdaptive
    c = 32
    mod = LFMResizeAdaptive(c, 7)
    shp = synth.shapes_of(mod)
    mod.load_state_dict(synth.synth_state_dict(shp, seed=4))
    x0 = synth.rand("lfm_x0", (3, c, 12, 20)).requires_grad_(True)
    y0, g0 = mod(x0)
    x1 = synth.rand("lfm_x1", (3, c, 6, 10)).requires_grad_(True)
    y1, g1 = mod(x1, g0)
    x2 = synth.rand("lfm_x2", (3, c, 3, 5)).requires_grad_(True)      # odd sizes: ifft2(s=(h,w)) path
    y2, g2 = mod(x2, g1)
    go = [synth.rand(f"lfm_go{i}", y.shape) for i, y in enumerate((y0, y1, y2))]
    loss = sum((y * g).sum() for y, g in zip((y0, y1, y2), go))
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(loss, [x0, x1, x2] + list(params.values()))
    arrays = dict(y0=y0, g0=g0, y1=y1, g1=g1, y2=y2, g2=g2, gx0=grads[0], gx1=grads[1], gx2=grads[2])
    for (k, _), g in zip(params.items(), grads[3:]):
        arrays["gp_" + k] = g
    save("lfm", {"c": c, "sigma": 7, "seed": 4, "param_shapes": shp}, **arrays)


def gen_fusion(name="fusion", nhead=8):
    ref_import.install()
    from models.segmentation import VisionLanguageFusionModule
    d = 64
    mod = VisionLanguageFusionModule(d_model=d, nhead=nhead)
    shp = synth.shapes_of(mod)
    mod.load_state_dict(synth.synth_state_dict(shp, seed=5))
    t, h, w, b, Lt = 2, 3, 5, 2, 6
    vis = synth.rand("fu_vis", (t, h, w, b, d)).requires_grad_(True)
    text = synth.rand("fu_text", (Lt, b, d)).requires_grad_(True)
    tpos = synth.rand("fu_tpos", (Lt, b, d))
    pm = torch.zeros(b, Lt, dtype=torch.bool)
    pm[1, 4:] = True
    out = mod(visual=vis, text=text, text_key_padding_mask=pm, text_pos=tpos, visual_pos=None)
    go = synth.rand("fu_go", out.shape)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad((out * go).sum(), [vis, text] + list(params.values()))
    arrays = dict(out=out, gvis=grads[0], gtext=grads[1], pad=pm)
    for (k, _), g in zip(params.items(), grads[2:]):
        arrays["gp_" + k] = g
    save(name, {"d": d, "nhead": nhead, "seed": 5, "t": t, "h": h, "w": w, "b": b, "Lt": Lt, "param_shapes": shp}, **arrays)


# ----------------------------------------------------------------------------------------------
def build_tiny(seed=1, B=2, **over):
    cfg = dict(TINY)
    cfg.update(over)
    args = ref_import.reference_args(**cfg)
    model, crit, _ = ref_import.build_reference_model(args, tiny_text(B))
    full = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=seed), strict=False)   # integer buffers keep their own values
    # bernoulli streams cannot be matched across implementations: every dropout (incl. the FeatureResizers'
    # hard-coded 0.1, ocpg.py:85-94) is disabled on both sides for the parity fixtures
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return args, cfg, model, crit, full


def gen_dynmask_mso():
    args, cfg, model, crit, _ = build_tiny()
    from util.misc import NestedTensor
    b, t, q, c, h, w = 2, 2, 3, 64, 6, 7
    feats = synth.rand("dm_feat", (b, t, c, h, w)).requires_grad_(True)
    n_par = model.num_gen_params
    params = (synth.rand("dm_par", (b, t * q, n_par)) * 0.2).requires_grad_(True)
    refp = synth.rand("dm_ref", (b, t * q, 2), uniform=True).requires_grad_(True)
    targets = [{"size": torch.tensor([h * 8, w * 8])}, {"size": torch.tensor([h * 8 - 5, w * 8 - 3])}]
    out = model.dynamic_mask_with_coords(feats, params, refp, targets)
    go = synth.rand("dm_go", out.shape)
    gf, gp, gr = torch.autograd.grad((out * go).sum(), (feats, params, refp))
    arrays = dict(dm_out=out, dm_gfeat=gf, dm_gpar=gp, dm_gref=gr)
    # MSO
    mso = model.mask_refine
    n = 3
    pm = synth.rand("mso_pm", (n, 16, 6, 7)).requires_grad_(True)
    f4 = synth.rand("mso_f4", (n, 256, 12, 14)).requires_grad_(True)
    f8 = synth.rand("mso_f8", (n, 512, 6, 7)).requires_grad_(True)
    o = mso(pm.clone(), [NestedTensor(f4, None), NestedTensor(f8, None)])
    go = synth.rand("mso_go", o.shape)
    mp = dict(mso.named_parameters())
    grads = torch.autograd.grad((o * go).sum(), [pm, f4, f8] + list(mp.values()))
    arrays.update(mso_out=o, mso_gpm=grads[0], mso_gf4=grads[1], mso_gf8=grads[2])
    for (k, _), g in zip(mp.items(), grads[3:]):
        arrays["mso_gp_" + k] = g
    save("dynmask_mso", {"cfg": cfg, "seed": 1, "b": b, "t": t, "q": q, "c": c, "h": h, "w": w, "n_par": n_par,
                         "sizes": [[h * 8, w * 8], [h * 8 - 5, w * 8 - 3]],
                         "mso_param_shapes": synth.shapes_of(mso)}, **arrays)


def gen_matcher_crit():
    args, cfg, model, crit, _ = build_tiny()
    b, t, q, H, W = 2, 2, 3, 64, 96
    targets = synth.synthetic_targets(b, t, H, W)
    targets[1]["valid"] = torch.tensor([1, 0])
    targets[1]["boxes"] = torch.tensor([[0.3, 0.4, 0.2, 0.3], [0.6, 0.5, 0.3, 0.2]])
    targets[0]["weights"] = synth.rand("mc_heat", (t, H, W), uniform=True) * targets[0]["masks"]
    outputs = {
        "pred_logits": synth.rand("mc_logits", (b, t, q, 1)).requires_grad_(True),
        "pred_boxes": (synth.rand("mc_boxes", (b, t, q, 4), uniform=True) * 0.5 + 0.2).requires_grad_(True),
        "pred_masks": synth.rand("mc_masks", (b, t, q, H // 2, W // 2)),
    }
    ind = model.matcher(outputs, targets)
    arrays = {"idx_src": torch.stack([i[0] for i in ind]), "idx_tgt": torch.stack([i[1] for i in ind])}
    # criterion on a synthetic "selected" output set
    out = {
        "pred_logits": outputs["pred_logits"], "pred_boxes": outputs["pred_boxes"],
        "pred_masks": synth.rand("mc_pm", (b, t, H, W)).requires_grad_(True),
        "pred_masks_low": synth.rand("mc_pml", (b, t, H // 2, W // 2)).requires_grad_(True),
        "ls_features": synth.rand("mc_ls", (b, t, 12, H // 2, W // 2)).requires_grad_(True),
        "frames": synth.rand("mc_fr", (b, t, 3, H // 2, W // 2)),
        "main_matcher_index": ind, "aux_matcher_index": [],
    }
    losses, src_m, tgt_m, weak_m = crit(out, targets)
    wd = crit.weight_dict
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    leaves = [out["pred_logits"], out["pred_boxes"], out["pred_masks"], out["pred_masks_low"], out["ls_features"]]
    grads = torch.autograd.grad(total, leaves)
    for k, v in losses.items():
        arrays["loss_" + k] = v
    arrays.update(total=total, g_logits=grads[0], g_boxes=grads[1], g_pm=grads[2], g_pml=grads[3], g_ls=grads[4],
                  src_m=src_m, tgt_m=tgt_m, weak_m=weak_m)
    save("matcher_crit", {"cfg": cfg, "b": b, "t": t, "q": q, "H": H, "W": W,
                          "weight_dict": {k: float(v) for k, v in wd.items()}}, **arrays)


def run_e2e(model, crit, B, T, H, W, sizes, train=True):
    from util.misc import NestedTensor
    clips = [synth.rand(f"e2e_clip{i}", (T, 3, h, w)) for i, (h, w) in enumerate(sizes)]
    x = torch.zeros(B, T, 3, H, W)
    mask = torch.ones(B, T, H, W, dtype=torch.bool)
    targets = []
    for i, (h, w) in enumerate(sizes):
        x[i, :, :, :h, :w] = clips[i]
        mask[i, :, :h, :w] = False
        targets.append(synth.synthetic_targets(1, T, h, w)[0])
    model.train(train)
    crit.train(train)
    out = model(NestedTensor(x, mask), ["a"] * B, targets)
    return out, targets


def gen_e2e(name, over, cases_, B, T, H, W, seed=1):
    """OCPG.forward + criterion + backward in train mode (+ the eval tail) for one configuration; `cases_` = [(tag, sizes)]."""
    arrays, meta = {}, {}
    for tag, sizes in cases_:
        args, cfg, model, crit, full = build_tiny(seed=seed, B=B, **over)
        out, targets = run_e2e(model, crit, B, T, H, W, sizes)
        losses, *_ = crit(out, targets)
        wd = crit.weight_dict
        total = sum(losses[k] * wd[k] for k in losses if k in wd)
        total.backward()
        for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low", "ls_features"):
            arrays[f"{tag}_{k}"] = out[k]
        arrays[f"{tag}_main_idx"] = torch.stack([i[0] for i in out["main_matcher_index"]])
        arrays[f"{tag}_aux_idx"] = torch.stack([torch.stack([i[0] for i in a]) for a in out["aux_matcher_index"]])
        for i, a in enumerate(out["aux_outputs"]):
            arrays[f"{tag}_aux{i}_pred_masks"] = a["pred_masks"]
            arrays[f"{tag}_aux{i}_pred_boxes"] = a["pred_boxes"]
        arrays[f"{tag}_total"] = total
        meta[f"{tag}_losses"] = {k: float(v) for k, v in losses.items()}
        gn = {k: (float(p.grad.norm()) if p.grad is not None else None) for k, p in model.named_parameters()}
        meta[f"{tag}_grad_norms"] = gn
        meta[f"{tag}_total_grad_norm"] = float(torch.norm(torch.stack([p.grad.norm() for p in model.parameters() if p.grad is not None])))
        named = dict(model.named_parameters())
        for k in ("query_embed.weight", "transformer.level_embed", "class_embed.1.weight", "controller.layers.2.bias",
                  "transformer.encoder.layers.0.self_attn.sampling_offsets.bias", "input_proj.3.0.bias", "input_proj.2.0.bias",
                  "backbone.0.body.layer2.0.conv1.weight", "mask_refine.out_conv.weight"):
            if k in named and named[k].grad is not None:
                arrays[f"{tag}_grad_{k}"] = named[k].grad
        meta[f"{tag}_sizes"] = sizes
        # eval tail
        args, cfg, model, crit, full = build_tiny(seed=seed, B=B, **over)
        with torch.no_grad():
            oute, _ = run_e2e(model, crit, B, T, H, W, sizes, train=False)
        for k in ("pred_logits", "pred_boxes", "pred_masks", "reference_points"):
            arrays[f"{tag}_eval_{k}"] = oute[k]
    meta.update(cfg=cfg, seed=seed, B=B, T=T, H=H, W=W, state_shapes=full, float_shapes=synth.shapes_of(model),
                weight_dict={k: float(v) for k, v in wd.items()})
    save(name, meta, **arrays)


def gen_e2e_tiny():
    gen_e2e("e2e_tiny", {}, (("nopad", [(192, 224), (192, 224)]), ("pad", [(192, 224), (160, 200)])), 2, 2, 192, 224)


def gen_e2e_d32():
    """Same step with head_dim 32 (hidden 64 / 2 heads): the channel count of every BASELINE configuration, i.e. the
    MSDeformAttn kernels the benchmark runs (column-tile scatter + gather-row backward, G = 8) sit inside the parity step."""
    gen_e2e("e2e_d32", dict(nheads=2), (("pad", [(192, 224), (160, 200)]),), 2, 2, 192, 224)


def gen_e2e_cfg1():
    """BASELINE config #1: ResNet-50, ONE frame, 256x256, 3 feature levels, 1 query (2 levels cannot run
    through OCPG.forward, ocpg.py:247); hidden 256 / 8 heads as in the launch scripts, 1 encoder + 1 decoder... no:
    the layer counts are kept small (1 + 2) to bound the fixture time, the shapes that make #1 special are all here."""
    gen_e2e("e2e_cfg1", dict(hidden_dim=256, mask_dim=256, dim_feedforward=512, nheads=8, num_feature_levels=3, num_queries=1,
                             num_frames=1), (("nopad", [(256, 256)]),), 1, 1, 256, 256)


def _install_engine_stubs():
    """engine.py imports evaluation-only packages at module level (cv2, pycocotools, datasets.*_eval): name stubs."""
    ref_import.install()
    ref_import._mod("cv2")
    ref_import._mod("pycocotools.coco", COCO=object)
    ref_import._mod("pycocotools.cocoeval", COCOeval=object)
    ds = ref_import._mod("datasets")
    ds.coco_eval = ref_import._mod("datasets.coco_eval", CocoEvaluator=object)
    ds.refexp_eval = ref_import._mod("datasets.refexp_eval", RefExpEvaluator=object)
    ds.a2d_eval = ref_import._mod("datasets.a2d_eval", calculate_precision_at_k_and_iou_metrics=None,
                                  calculate_bbox_precision_at_k_and_iou_metrics=None)


def gen_train_step():
    """Rows f1 / f2: ONE iteration of the reference's own engine.train_one_epoch (engine.py:29-123) on the tiny model with
    the optimizer of main.py:76-99 (AdamW, four name-based LR groups, weight decay 5e-4, clip 0.1), then a checkpoint
    written by the reference's util.misc.save_on_master with the dict of main.py:229-236."""
    import argparse
    import tempfile
    _install_engine_stubs()
    import engine
    import util.misc as utils
    from util.misc import NestedTensor
    args, cfg, model, crit, full = build_tiny()
    B, T, H, W = 2, 2, 192, 224
    sizes = [(192, 224), (160, 200)]
    x = torch.zeros(B, T, 3, H, W)
    mask = torch.ones(B, T, H, W, dtype=torch.bool)
    targets = []
    for i, (h, w) in enumerate(sizes):
        x[i, :, :, :h, :w] = synth.rand(f"e2e_clip{i}", (T, 3, h, w))
        mask[i, :, :h, :w] = False
        t = synth.synthetic_targets(1, T, h, w)[0]
        t["caption"] = "a"
        targets.append(t)

    def match(n, kws):
        return any(k in n for k in kws)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [      # main.py:76-96
        {"params": [p for n, p in named if not match(n, args.lr_backbone_names) and not match(n, args.lr_text_encoder_names)
                    and not match(n, args.lr_linear_proj_names)], "lr": args.lr},
        {"params": [p for n, p in named if match(n, args.lr_backbone_names)], "lr": args.lr_backbone},
        {"params": [p for n, p in named if match(n, args.lr_text_encoder_names)], "lr": args.lr_text_encoder},
        {"params": [p for n, p in named if match(n, args.lr_linear_proj_names)], "lr": args.lr * args.lr_linear_proj_mult},
    ]
    opt = torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [3, 5])
    scaler = torch.amp.GradScaler("cpu", enabled=False)
    args.amp = False
    args.output_dir = tempfile.mkdtemp()
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    stats, _ = engine.train_one_epoch(args, model, crit, [(NestedTensor(x, mask), targets)], opt, scaler, torch.device("cpu"), 0,
                                      max_norm=args.clip_max_norm, lr_scheduler=sched)
    after = dict(model.named_parameters())
    meta = {"cfg": cfg, "seed": 1, "B": B, "T": T, "H": H, "W": W, "sizes": sizes, "float_shapes": synth.shapes_of(model),
            "loss": float(stats["loss"]), "grad_norm": float(stats["grad_norm"]), "lr": float(stats["lr"]),
            "clip_max_norm": float(args.clip_max_norm), "weight_decay": float(args.weight_decay),
            "group_lrs": [g["lr"] for g in opt.param_groups], "group_sizes": [len(g["params"]) for g in opt.param_groups],
            "group_first_names": [next((n for n, p in named if p is g["params"][0]), None) if g["params"] else None for g in opt.param_groups],
            "param_abs_sum": {k: float(v.detach().double().abs().sum()) for k, v in after.items()},
            "param_delta_norm": {k: float((v.detach() - before[k]).double().norm()) for k, v in after.items()}}
    arrays = {}
    for k in ("query_embed.weight", "transformer.level_embed", "class_embed.1.weight", "controller.layers.2.bias",
              "transformer.encoder.layers.0.self_attn.sampling_offsets.bias", "input_proj.3.0.bias", "mask_refine.out_conv.weight",
              "bbox_embed.0.layers.2.bias", "text_proj.fc.bias"):
        if k in after:
            arrays["after_" + k] = after[k].detach()
            arrays["delta_" + k] = after[k].detach() - before[k]
    save("train_step", meta, **arrays)

    # ---- f2: a checkpoint written by the reference (main.py:226-236 -> util/misc.py:444-446), small enough to commit:
    # the same dict, restricted to the head modules (the ResNet body alone is 94 MB); optimizer rebuilt over exactly those
    # parameters and stepped once so that its state_dict carries real exp_avg / exp_avg_sq / step entries.
    sched.step()
    small = ("query_embed.", "class_embed.", "bbox_embed.", "controller.", "mask_refine.", "ls_feat_viz.", "ls_text_proj.",
             "transformer.level_embed", "transformer.reference_points.", "sentence_proj.")
    sub = [(n, p) for n, p in model.named_parameters() if n.startswith(small) and not n.startswith("transformer.decoder.bbox_embed")]
    opt2 = torch.optim.AdamW([{"params": [p for n, p in sub if "reference_points" not in n], "lr": args.lr},
                              {"params": [p for n, p in sub if "reference_points" in n], "lr": args.lr * args.lr_linear_proj_mult}],
                             lr=args.lr, weight_decay=args.weight_decay)
    sched2 = torch.optim.lr_scheduler.MultiStepLR(opt2, [1, 5])
    for _, p in sub:
        p.grad = synth.rand("ck_grad_" + _, p.shape) * 1e-3
    opt2.step()
    sched2.step()
    sched2.step()                      # last_epoch = 2: one milestone (epoch 1) passed -> lr * 0.1
    path = os.path.join(HERE, "ckpt_ref.pth")
    ns = argparse.Namespace(**{k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, list, tuple, type(None)))})
    ns.output_dir = "output"
    utils.save_on_master({"model": {n: p.detach().clone() for n, p in sub}, "optimizer": opt2.state_dict(),
                          "lr_scheduler": sched2.state_dict(), "epoch": 1, "args": ns, "grad_scaler": scaler.state_dict()}, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    meta2 = {"keys": [n for n, _ in sub], "shapes": {n: list(p.shape) for n, p in sub}, "epoch": 1,
             "lrs_after_resume": [g["lr"] for g in opt2.param_groups], "last_epoch": sched2.last_epoch,
             "abs_sum": {n: float(p.detach().double().abs().sum()) for n, p in sub},
             "exp_avg_abs_sum": [float(opt2.state[p]["exp_avg"].double().abs().sum()) for _, p in sub]}
    save("ckpt_ref_manifest", meta2)


# head_dim is 32 at every stage (like Swin-T/S/B), so the product's fused HIP window-attention kernel is what runs on the GPU
SWIN_TINY = dict(patch_size=(1, 4, 4), embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=(8, 7, 7),
                 mlp_ratio=2.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                 patch_norm=True, use_checkpoint=False)


def gen_swin3d():
    ref_import.install()
    import models.video_swin_transformer as vs
    arrays, meta = {}, {"swin_tiny": SWIN_TINY}
    # (a) WindowAttention3D, small window, with and without a shift mask
    wa = vs.WindowAttention3D(24, (2, 3, 3), 3, qkv_bias=True)
    shp = synth.shapes_of(wa)
    wa.load_state_dict(synth.synth_state_dict(shp, seed=6), strict=False)
    meta["wa_shapes"] = shp
    x = synth.rand("wa_x", (8, 18, 24)).requires_grad_(True)
    mask = vs.compute_mask(2, 6, 6, (2, 3, 3), (0, 1, 1), "cpu")
    arrays["wa_mask"] = mask
    for tag, m in (("nomask", None), ("mask", mask)):
        y = wa(x, m)
        g = torch.autograd.grad((y * synth.rand("wa_go", y.shape)).sum(), [x] + list(wa.parameters()))
        arrays[f"wa_{tag}_y"], arrays[f"wa_{tag}_gx"] = y, g[0]
        for (k, _), gg in zip(wa.named_parameters(), g[1:]):
            arrays[f"wa_{tag}_gp_{k}"] = gg
    # (a') the real clamped case: module built for (8,7,7), tokens of a (5,7,7) window -> index table sliced [:245,:245]
    wb = vs.WindowAttention3D(24, (8, 7, 7), 3, qkv_bias=True)
    shpb = synth.shapes_of(wb)
    wb.load_state_dict(synth.synth_state_dict(shpb, seed=7), strict=False)
    meta["wb_shapes"] = shpb
    xb = synth.rand("wb_x", (2, 245, 24))
    arrays["wb_y"] = wb(xb, None)
    # (a'') head_dim 32 (the fused-kernel shape): dim 96 / 3 heads, clamped (5,7,7) window, real shift mask, with grads
    wc = vs.WindowAttention3D(96, (8, 7, 7), 3, qkv_bias=True)
    shpw = synth.shapes_of(wc)
    wc.load_state_dict(synth.synth_state_dict(shpw, seed=11), strict=False)
    meta["wc_shapes"] = shpw
    maskc = vs.compute_mask(5, 14, 14, (5, 7, 7), (0, 3, 3), "cpu")          # 4 windows
    xw = synth.rand("wc_x", (8, 245, 96)).requires_grad_(True)
    for tag, m in (("nomask", None), ("mask", maskc)):
        y = wc(xw, m)
        g = torch.autograd.grad((y * synth.rand("wc_go", y.shape)).sum(), [xw] + list(wc.parameters()))
        arrays[f"wc_{tag}_y"], arrays[f"wc_{tag}_gx"] = y, g[0]
        for (k, _), gg in zip(wc.named_parameters(), g[1:]):
            if "bias_table" in k:
                arrays[f"wc_{tag}_gtable"] = gg
            meta.setdefault(f"wc_{tag}_grad_norms", {})[k] = float(gg.norm())
    # (b) shift masks
    arrays["mask_5_14_21"] = vs.compute_mask(5, 14, 21, (5, 7, 7), (0, 3, 3), "cpu")
    arrays["mask_8_7_14"] = vs.compute_mask(8, 7, 14, (4, 7, 7), (2, 0, 3), "cpu")
    # (c) a shifted block on a map that needs padding; temporal axis clamped
    blk = vs.SwinTransformerBlock3D(24, 3, window_size=(8, 7, 7), shift_size=(4, 3, 3), mlp_ratio=2.0)
    shpc = synth.shapes_of(blk)
    blk.load_state_dict(synth.synth_state_dict(shpc, seed=8), strict=False)
    meta["blk_shapes"] = shpc
    xc = synth.rand("blk_x", (2, 5, 10, 13, 24)).requires_grad_(True)
    am = vs.compute_mask(5, 14, 14, (5, 7, 7), (0, 3, 3), "cpu")
    yc = blk(xc, am)
    gc = torch.autograd.grad((yc * synth.rand("blk_go", yc.shape)).sum(), [xc] + list(blk.parameters()))
    arrays["blk_y"], arrays["blk_gx"] = yc, gc[0]
    meta["blk_grad_norms"] = {k: float(g.norm()) for (k, _), g in zip(blk.named_parameters(), gc[1:])}
    # (d) patch merging with odd sizes
    pm = vs.PatchMerging(24)
    shpd = synth.shapes_of(pm)
    pm.load_state_dict(synth.synth_state_dict(shpd, seed=9))
    meta["pm_shapes"] = shpd
    arrays["pm_y"] = pm(synth.rand("pm_x", (2, 3, 5, 7, 24)))
    # (e) the whole backbone wrapper, tiny configuration, 5 frames of 64 x 96
    bb = vs.VideoSwinTransformerBackbone(False, None, True, (0, 1, 2, 3), **SWIN_TINY)
    shpe = synth.shapes_of(bb)
    bb.load_state_dict(synth.synth_state_dict(shpe, seed=10), strict=False)
    meta["bb_shapes"] = shpe
    xe = synth.rand("bb_x", (5, 3, 64, 96)).requires_grad_(True)
    out = bb(xe, 5)
    loss = 0
    for k, v in out.items():
        arrays[f"bb_out{k}"] = v
        loss = loss + (v * synth.rand(f"bb_go{k}", v.shape)).sum()
    ge = torch.autograd.grad(loss, [xe] + list(bb.parameters()), allow_unused=True)
    arrays["bb_gx"] = ge[0]
    meta["bb_grad_norms"] = {k: (float(g.norm()) if g is not None else None) for (k, _), g in zip(bb.named_parameters(), ge[1:])}
    save("swin3d", meta, **arrays)


def gen_swin_n392():
    """BASELINE config #5's attention shape: the FULL (8,7,7) window (N = 392 tokens, 8 frames >= window depth so no clamping,
    video_swin_transformer.py:71-84), head_dim 32, with the shift mask of a (0,3,3) shift (compute_mask :316-329)."""
    ref_import.install()
    import models.video_swin_transformer as vs
    wa = vs.WindowAttention3D(64, (8, 7, 7), 2, qkv_bias=True)
    shp = synth.shapes_of(wa)
    wa.load_state_dict(synth.synth_state_dict(shp, seed=13), strict=False)
    mask = vs.compute_mask(8, 14, 14, (8, 7, 7), (0, 3, 3), "cpu")            # 4 windows
    x = synth.rand("w392_x", (4, 392, 64)).requires_grad_(True)
    arrays, meta = {}, {"shapes": shp}
    for tag, m in (("nomask", None), ("mask", mask)):
        y = wa(x, m)
        g = torch.autograd.grad((y * synth.rand("w392_go", y.shape)).sum(), [x] + list(wa.parameters()))
        arrays[f"{tag}_y"], arrays[f"{tag}_gx"] = y, g[0]
        meta[f"{tag}_grad_norms"] = {k: float(gg.norm()) for (k, _), gg in zip(wa.named_parameters(), g[1:])}
        arrays[f"{tag}_gtable"] = dict(zip([k for k, _ in wa.named_parameters()], g[1:]))["relative_position_bias_table"]
    save("swin_n392", meta, **arrays)


def gen_e2e_swin():
    """OCPG end to end with the Video-Swin backbone (tiny Swin configuration), train mode."""
    ref_import.install()
    import models.video_swin_transformer as vs
    vs.configs["video_swin_t_p4w7"] = dict(SWIN_TINY)
    arrays, meta = {}, {}
    args, cfg, model, crit, full = build_tiny(backbone="video_swin_t_p4w7", backbone_pretrained=None, use_checkpoint=False, output_levels=4)
    B, T, H, W = 2, 2, 192, 224
    sizes = [(192, 224), (160, 200)]
    out, targets = run_e2e(model, crit, B, T, H, W, sizes)
    losses, *_ = crit(out, targets)
    wd = crit.weight_dict
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low"):
        arrays[f"pad_{k}"] = out[k]
    arrays["pad_main_idx"] = torch.stack([i[0] for i in out["main_matcher_index"]])
    arrays["pad_aux_idx"] = torch.stack([torch.stack([i[0] for i in a]) for a in out["aux_matcher_index"]])
    arrays["pad_total"] = total
    meta["pad_losses"] = {k: float(v) for k, v in losses.items()}
    meta["pad_grad_norms"] = {k: (float(p.grad.norm()) if p.grad is not None else None) for k, p in model.named_parameters()}
    meta["pad_sizes"] = sizes
    meta.update(cfg=cfg, swin_cfg=SWIN_TINY, seed=1, B=B, T=T, H=H, W=W, state_shapes=full, float_shapes=synth.shapes_of(model))
    save("e2e_swin", meta, **arrays)


def gen_clip_transforms():
    """Input side (row f4): the reference's OWN target arithmetic of datasets/transforms_video.py (resize incl. its size rule, crop,
    Check, hflip + caption swap, Normalize) and datasets/ytvos.py:22-38 (weight2mask) on seeded synthetic clips.  Those modules
    import torchvision / cv2 / h5py at module level; the name stubs below add the five torchvision.transforms.functional calls the
    image side needs as their PIL one-liners (what torchvision itself does for PIL inputs) -- the image pixels in this fixture are
    therefore PIL's, the target tensors are the reference's."""
    import random as pyrandom

    from PIL import Image
    ref_import.install()
    tv = sys.modules["torchvision"]

    def to_tensor(im):
        return torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float() / 255.0

    def normalize(x, mean, std):
        return (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)

    tvf = ref_import._mod("torchvision.transforms.functional",
                          crop=lambda im, i, j, h, w: im.crop((j, i, j + w, i + h)),
                          hflip=lambda im: im.transpose(Image.FLIP_LEFT_RIGHT),
                          vflip=lambda im: im.transpose(Image.FLIP_TOP_BOTTOM),
                          resize=lambda im, size: im.resize((size[1], size[0]), Image.BILINEAR),
                          to_tensor=to_tensor, normalize=normalize, pad=None)
    tv.transforms.functional = tvf
    tv.transforms.RandomCrop = type("RandomCrop", (), {})
    tv.transforms.RandomErasing = type("RandomErasing", (), {})
    for name in ("cv2", "h5py"):
        if name not in sys.modules:
            ref_import._mod(name)
    import importlib.machinery
    import types
    pkg = types.ModuleType("datasets")           # bypass datasets/__init__.py (it imports every dataset: torchvision.io, pycocotools, ...)
    pkg.__path__ = [os.path.join(ref_import.REF, "datasets")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("datasets", None, is_package=True)
    sys.modules["datasets"] = pkg
    import datasets.transforms_video as RT
    import datasets.ytvos as RY

    g = torch.Generator().manual_seed(11)
    T_, H, W = 3, 96, 160
    frames = torch.randint(0, 256, (T_, H, W, 3), dtype=torch.uint8, generator=g)
    frames = (frames.float() * 0.25 + torch.linspace(0, 190, W)[None, None, :, None]).to(torch.uint8)      # texture on a ramp
    clip = [Image.fromarray(f.numpy()) for f in frames]
    masks = torch.zeros(T_, H, W)
    masks[0, 20:50, 30:90] = 1
    masks[1, 40:70, 100:150] = 1                      # frame 2 has no object
    boxes = torch.tensor([[30.0, 20, 89, 49], [100, 40, 149, 69], [0, 0, 0, 0]])
    target = {"frames_idx": torch.arange(T_), "labels": torch.full((T_,), 4), "boxes": boxes, "masks": masks,
              "valid": torch.tensor([1, 1, 0]), "caption": "the left zebra, right of the upright lefty", "orig_size": torch.as_tensor([H, W]),
              "size": torch.as_tensor([H, W]), "weights": torch.rand(T_, H, W, generator=g), "weak_masks": torch.rand(T_, H, W, generator=g)}
    arrays = {"frames": frames, "masks": masks, "boxes": boxes, "weights": target["weights"], "weak_masks": target["weak_masks"]}
    meta = {"caption": target["caption"], "cases": {}}

    def pack(tag, imgs, t):
        if imgs is not None:       # PIL frames are uint8: stored as such (the normalised clip stays float)
            arrays[tag + "_img"] = torch.stack([im if torch.is_tensor(im) else torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1) for im in imgs])
        for k in ("boxes", "masks", "valid", "size", "weights", "weak_masks", "area"):
            if k in t and (k not in ("weights", "weak_masks") or tag in ("rs72", "crop_in", "flip")):
                arrays[f"{tag}_{k}"] = t[k].to(torch.uint8) if (t[k].dtype == torch.bool or k == "masks") else t[k]

    resize_cases = {"rs72": (72, 100), "rs64": (64, None), "rs128": (128, 200), "rs_pair": ((50, 30), None), "rs_same": (96, None)}
    for tag, (size, max_size) in resize_cases.items():
        imgs, t = RT.resize(clip, dict(target), size, max_size)
        pack(tag, imgs, t)
        meta["cases"][tag] = {"size": size, "max_size": max_size, "out_hw": [imgs[0].size[1], imgs[0].size[0]]}
    # the size rule alone on shapes of the real datasets (w, h order as PIL reports it)
    rule = []
    for (h, w, size, max_size) in [(480, 854, 360, 640), (360, 640, 360, 640), (720, 1280, 512, 640), (720, 1280, 288, 640), (500, 300, 400, None),
                                   (300, 500, 600, None), (1080, 1920, 448, 640), (640, 480, 480, 640), (333, 777, 392, 640), (405, 720, 360, 640)]:
        probe = [Image.new("RGB", (w, h))]
        out, _ = RT.resize(probe, None, size, max_size)
        rule.append([h, w, size, -1 if max_size is None else max_size, out[0].size[1], out[0].size[0]])
    arrays["size_rule"] = torch.tensor(rule)
    crop_cases = {"crop_in": (10, 20, 60, 90), "crop_miss": (60, 0, 30, 25), "crop_edge": (30, 60, 66, 100)}
    for tag, region in crop_cases.items():
        imgs, t = RT.crop(clip, dict(target), region)
        imgs, t = RT.Check()(imgs, t)
        pack(tag, imgs, t)
        meta["cases"][tag] = {"region": list(region)}
    imgs, t = RT.hflip(clip, dict(target))
    pack("flip", imgs, t)
    pyrandom.seed(0)
    _, t = RT.RandomHorizontalFlip(p=1.0)(clip, dict(target))
    meta["flipped_caption"] = t["caption"]
    tens, _ = RT.ToTensor()(clip, dict(target))
    imgs, t = RT.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])(tens, dict(target))
    pack("norm", imgs, t)
    # weak supervision: heat maps -> instance mask + box (ytvos.py:22-38)
    heat = torch.rand(3, 40, 64, generator=g) * 0.45
    heat[0, 5:20, 10:30] += 0.5
    heat[1, 15:35, 25:60] += 0.5
    heat[2, 0:4, 0:4] += 0.1                                   # never beats the 0.5 background plane
    arrays["heat"] = heat
    for k in range(3):
        m, b = RY.weight2mask(heat, k)
        arrays[f"w2m_mask{k}"], arrays[f"w2m_box{k}"] = m, b
    save("clip_transforms", meta, **arrays)


def _ref_lines(path, start_marker, end_marker):
    """The statements of a reference script between two marker lines (inclusive of the end line), dedented -- read from the
    reference checkout at GENERATION time and executed here; no reference text is written anywhere."""
    import textwrap
    lines = open(os.path.join(ref_import.REF, path)).read().split("\n")
    a = next(i for i, ln in enumerate(lines) if start_marker in ln)
    b = next(i for i, ln in enumerate(lines) if end_marker in ln and i > a)
    return textwrap.dedent("\n".join(lines[a:b + 1]))


def gen_infer_collate():
    """Rows f3 / f4 of SURVEY section 8, pinned to the reference's OWN statements:
      collate_*   util.misc.collate_fn / nested_tensor_from_videos_list / nested_tensor_from_tensor_list (util/misc.py:299-379) on
                  ragged clips;
      infer_*     the per-video core of inference_davis.py (:203-250: clip chopping, model call on a frame list, best-query
                  selection, un-pad, resize to the original size, sigmoid, concatenation) and its multi-object merge (:254-261),
                  EXECUTED from the reference file on JPEG frames written to a temporary folder, with the tiny reference model."""
    import contextlib
    import tempfile
    import types
    from PIL import Image
    ref_import.install()
    import util.misc as rmisc
    arrays, meta = {}, {}
    # ---- f4: collate ----------------------------------------------------------------------------------------------------
    sizes = [(2, 3, 20, 30), (3, 3, 33, 17), (1, 3, 32, 64)]
    clips = [synth.rand(f"col_clip{i}", s) for i, s in enumerate(sizes)]
    targets = [{"k": i} for i in range(len(clips))]
    samples, tg = rmisc.collate_fn(list(zip(clips, targets)))
    assert [t["k"] for t in tg] == [0, 1, 2]
    arrays["collate_tensors"], arrays["collate_mask"] = samples.tensors, samples.mask
    flat = [synth.rand(f"col_img{i}", s) for i, s in enumerate([(6, 10, 12), (3, 9, 14), (9, 11, 5)])]       # [T*3, h, w] stacks
    nt = rmisc.nested_tensor_from_tensor_list(flat, size_divisibility=8, split=True)
    arrays["split_tensors"], arrays["split_mask"] = nt.tensors, nt.mask
    nt1 = rmisc.nested_tensor_from_videos_list(clips[:2], size_divisibility=1)
    arrays["nodiv_tensors"], arrays["nodiv_mask"] = nt1.tensors, nt1.mask
    meta["collate_sizes"] = sizes
    meta["split_sizes"] = [(6, 10, 12), (3, 9, 14), (9, 11, 5)]
    # ---- f3: inference core -------------------------------------------------------------------------------------------
    args, cfg, model, crit, full = build_tiny(B=1, dataset_file="davis")
    model.eval()
    g = torch.Generator().manual_seed(123)
    video_len, crop_len, oh, ow, th, tw = 5, 2, 320, 400, 160, 200
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def transform(img):         # the caller's side (torchvision Resize + ToTensor + Normalize in the reference): PIL bilinear here
        img = img.resize((tw, th), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255.0
        return (x - mean) / std
    objs_logits, objs_masks, inputs = [], [], None
    with tempfile.TemporaryDirectory() as img_folder:
        video_name = "vid"
        os.makedirs(os.path.join(img_folder, video_name))
        frames = ["%05d" % i for i in range(video_len)]
        for f in frames:
            px = (torch.rand(oh // 8, ow // 8, 3, generator=g) * 255).to(torch.uint8).numpy()
            Image.fromarray(px).resize((ow, oh), Image.BICUBIC).save(os.path.join(img_folder, video_name, f + ".jpg"), quality=95)
        inputs = torch.stack([transform(Image.open(os.path.join(img_folder, video_name, f + ".jpg")).convert("RGB")) for f in frames])
        clip_loop = _ref_lines("inference_davis.py", "# 3. for each clip", "all_pred_masks = torch.cat(all_pred_masks, dim=0)")
        for obj_id in range(2):         # two "objects" = two expressions (two primed text stand-ins)
            tf, ts, tm = tiny_text(1)
            model.text_encoder.primed = (tf * (1.0 + 0.5 * obj_id), ts * (1.0 - 0.3 * obj_id), tm)
            ns = dict(video_len=video_len, crop_len=crop_len, frames=frames, img_folder=img_folder, video_name=video_name, transform=transform,
                      args=types.SimpleNamespace(device="cpu", amp=False), model=model, exp="a thing", obj_id=obj_id,
                      autocast=lambda enabled: contextlib.nullcontext(), torch=torch, F=torch.nn.functional, os=os, Image=Image,
                      all_pred_logits=[], all_pred_masks=[])
            exec(compile(clip_loop, "inference_davis.py[clip loop]", "exec"), ns)
            objs_logits.append(ns["all_pred_logits"])
            objs_masks.append(ns["all_pred_masks"])
            arrays[f"infer_logits{obj_id}"], arrays[f"infer_masks{obj_id}"] = ns["all_pred_logits"], ns["all_pred_masks"]
    merge = _ref_lines("inference_davis.py", "# handle a complete image", "out_masks = torch.argmax(anno_masks, dim=0)")
    ns = dict(anno_logits=objs_logits, anno_masks=[m.clone() for m in objs_masks], torch=torch, args=types.SimpleNamespace(device="cpu"))
    exec(compile(merge, "inference_davis.py[merge]", "exec"), ns)
    arrays["infer_labels"] = ns["out_masks"].to(torch.uint8)
    arrays["infer_frames"] = inputs
    meta.update(cfg=cfg, seed=1, state_shapes=full, float_shapes=synth.shapes_of(model), video_len=video_len, crop_len=crop_len,
                origin=[oh, ow], frame_size=[th, tw], text_scale=[[1.0, 1.0], [1.5, 0.7]])
    save("infer_collate", meta, **arrays)


GENS = {"msda_testpy": gen_msda_testpy, "msda_cases": gen_msda_cases, "msda_module": gen_msda_module,
        "transformer": gen_transformer, "lfm": gen_lfm, "fusion": gen_fusion, "dynmask_mso": gen_dynmask_mso,
        "matcher_crit": gen_matcher_crit, "e2e_tiny": gen_e2e_tiny, "e2e_d32": gen_e2e_d32, "e2e_cfg1": gen_e2e_cfg1,
        "train_step": gen_train_step,
        # head_dim 32 (d 64 / 2 heads): the module-level reference vectors reach the production kernels (msda_*<8>, attn_smallk)
        "msda_module_d32": lambda: gen_msda_module("msda_module_d32", M=2),
        "transformer_d32": lambda: gen_transformer("transformer_d32", nhead=2),
        "fusion_d32": lambda: gen_fusion("fusion_d32", nhead=2),
        "infer_collate": gen_infer_collate,
        "swin3d": gen_swin3d, "swin_n392": gen_swin_n392, "e2e_swin": gen_e2e_swin, "clip_transforms": gen_clip_transforms}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GENS)
    for n in names:
        GENS[n]()
