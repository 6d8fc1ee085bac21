"""BASELINE config #5 end to end at a reduced size, on the GPU: Video-Swin backbone (head_dim 32 at every stage -> the fused
HIP window attention, matrix-core variant under fp16) + captions through the REAL TextEncoder (HF RoBERTa-base, random init:
`checkpoints/roberta-base` is absent) + fp16 autocast + GradScaler (engine.py:98-106) + a clip width that
collate pads (8 x 160 x 172 -> 160 x 192, util/misc.py:299-307: padding masks, valid ratios, masked levels).

Checker (test infrastructure, CPU fp32): oracle/ocpg_ref.py for everything after the backbone, fed with
  * the text features of the SAME RoBERTa weights evaluated by HF on the CPU -- RoBERTa arithmetic is third-party and
    parity-UNPINNED (stated in DESIGN.md section 2): this pins GPU-vs-CPU agreement of the path, not HF itself;
  * the backbone maps of the SAME Swin weights evaluated by the product's own generic (tensor-op) path on the CPU, which is
    what tests/test_swin_cpu.py pins to the reference's vectors (swin3d.npz, e2e_swin.npz).
Bounds: fp32 product vs oracle = the usual 1e-3 on mask logits; fp16 product vs fp32 product = stated per output below.
"""
import copy
import os
import sys

import pytest
import torch

import cases
import synth

pytestmark = pytest.mark.gpu

SWIN_TINY = dict(patch_size=(1, 4, 4), embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=(8, 7, 7), mlp_ratio=2.0,
                 qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, patch_norm=True, use_checkpoint=False)
CAPTIONS = ["the person on the left riding a red bike", "a small dog"]          # different lengths: text padding mask


def _build(dev):
    from ocpg_amd.models import build_model
    args = cases.default_args(device=str(dev), backbone="video_swin_t_p4w7", video_swin_cfg=SWIN_TINY, hidden_dim=256, mask_dim=256,
                              nheads=8, dim_feedforward=256, enc_layers=1, dec_layers=2, num_frames=8, num_queries=3,
                              num_feature_levels=4, dropout=0.0, text_encoder_lazy=False, freeze_text_encoder=True, amp=True)
    torch.manual_seed(0)
    model, crit, _ = build_model(args)
    # seeded fan-in-scaled values for everything except the (HF-initialised, frozen) RoBERTa and the integer buffers
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if v.dtype.is_floating_point and not k.startswith("text_encoder.")}
    model.load_state_dict(synth.synth_state_dict(shapes, seed=21), strict=False)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.text_encoder.text_backbone.eval()         # HF dropout off: the frozen encoder is deterministic on both sides
    return args, model, crit


def _batch(dev):
    from ocpg_amd.util.misc import collate_fn
    T, H, W = 8, 160, 172
    clips, targets = [], []
    for i in range(2):
        clips.append(synth.rand(f"cfg5_clip{i}", (T, 3, H, W)))
        t = synth.synthetic_targets(1, T, H, W)[0]
        t["caption"] = CAPTIONS[i]
        targets.append(t)
    samples, targets = collate_fn(list(zip(clips, targets)))
    assert tuple(samples.tensors.shape) == (2, T, 3, 160, 192) and bool(samples.mask[..., 172:].all()) and not bool(samples.mask[..., :172].any())
    return samples, list(targets)


def _to(dev, samples, targets):
    from ocpg_amd.util.misc import NestedTensor
    m = samples.mask.to(dev)
    if hasattr(samples.mask, "_ocpg_key"):
        m._ocpg_key = samples.mask._ocpg_key
    return NestedTensor(samples.tensors.to(dev), m), [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]


def test_config5_swin_roberta_fp16_gradscaler_step(dev, monkeypatch):
    from oracle import ocpg_ref
    from ocpg_amd import _lib, engine
    from ocpg_amd.models import fallbacks
    from ocpg_amd.util.misc import NestedTensor
    monkeypatch.setenv("OCPG_STRICT_HIP", "1")          # no library attention fallback anywhere in this configuration
    fallbacks.reset()
    args, model, crit = _build(dev)
    cpu_model = copy.deepcopy(model)                    # the checker's backbone + text encoder (CPU, fp32, generic tensor-op paths)
    model.to(dev), crit.to(dev)
    model.train(), crit.train()
    model.text_encoder.text_backbone.eval()
    samples, targets = _batch(dev)

    # ---- checker: CPU fp32 -------------------------------------------------------------------------------------------
    with torch.no_grad():
        tf, ts, tpad = cpu_model.text_encoder(CAPTIONS, torch.device("cpu"))
        assert tpad.dtype == torch.bool and bool(tpad[1].any()) and not bool(tpad[0].any())
        cpu_model.eval()
        feats = [nt.tensors for nt in cpu_model.backbone[0](NestedTensor(samples.tensors.flatten(0, 1).clone(), samples.mask.flatten(0, 1).clone()),
                                                            num_frames=8).values()]
    P = {k: v.detach().clone() for k, v in cpu_model.state_dict().items() if not k.startswith("text_encoder.")}
    ref_out, ref_losses, ref_total = ocpg_ref.train_step_loss(P, ocpg_ref.cfg_from_args(args), samples.tensors, samples.mask, (tf, ts, tpad),
                                                              targets, features=feats)

    # ---- product, fp32 on the GPU: captions through TextEncoder -> RoBERTa on the GPU ----------------------------------
    calls = _lib.census(True)
    s_dev, t_dev = _to(dev, samples, targets)
    out32 = model(s_dev, CAPTIONS, t_dev)
    losses32, *_ = crit(out32, t_dev)
    total32 = crit.weighted_sum(losses32)
    assert torch.equal(torch.cat([i[0] for i in out32["main_matcher_index"]]).cpu(), ref_out["main_idx"])
    for k, tol in (("pred_logits", 2e-4), ("pred_boxes", 1e-4), ("pred_masks", 1e-3), ("pred_masks_low", 1e-3)):
        err = (out32[k].detach().cpu() - ref_out[k]).abs().max().item()
        assert err <= tol + 2e-5 * ref_out[k].abs().max().item(), (k, err)
    assert abs(total32.item() - ref_total.item()) <= 1e-3 * abs(ref_total.item()), (total32.item(), ref_total.item())
    assert calls.get("ocpg_win_attn_fwd", 0) >= 8, calls          # every Swin block through the fused window kernel
    total32.backward()
    norm32 = engine.total_grad_norm([p for p in model.parameters() if p.requires_grad])
    model.zero_grad(set_to_none=True)

    # ---- product, fp16 autocast + GradScaler: the reference's --amp step (engine.py:98-106) ----------------------------
    import bench
    crit.iter = 0
    opt = bench.make_optimizer(model, args, fused=False)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    before = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
    calls = _lib.census(True)
    s_dev, t_dev = _to(dev, samples, targets)
    with torch.autocast("cuda", dtype=torch.float16):
        out16 = model(s_dev, CAPTIONS, t_dev)
    assert calls.get("ocpg_win_attn_fwd", 0) >= 8, calls          # (fp16 storage: the library routes it to the matrix-core kernel)
    assert torch.equal(torch.cat([i[0] for i in out16["main_matcher_index"]]).cpu(), ref_out["main_idx"])
    # fp16 storage of activations (10-bit mantissa) through backbone + neck + 1 + 2 transformer layers: bounds relative to each
    # output's range, against the fp32 product run above
    for k, rel in (("pred_logits", 2e-2), ("pred_boxes", 1e-2), ("pred_masks", 3e-2), ("pred_masks_low", 3e-2)):
        a, b = out16[k].detach().float(), out32[k].detach().float()
        err = (a - b).abs().max().item()
        assert err <= rel * b.abs().max().item() + 1e-3, (k, err, b.abs().max().item())
    del out16
    crit.iter = 0
    s_dev, t_dev = _to(dev, samples, targets)
    loss_value, loss_dict, norm = engine.train_step(model, crit, s_dev, CAPTIONS, t_dev, opt, max_norm=args.clip_max_norm,
                                                    amp_dtype=torch.float16, grad_scaler=scaler)
    _lib.census(False)
    assert calls.get("ocpg_win_attn_bwd_mfma", 0) >= 8, calls     # fp16 storage -> the matrix-core window backward
    assert abs(loss_value - ref_total.item()) <= 2e-2 * abs(ref_total.item()), (loss_value, ref_total.item())
    assert torch.isfinite(norm), norm                                      # unscaled gradients finite -> the scaler took the step
    assert scaler.get_scale() == 1024.0                                     # no overflow: scale kept (growth interval not reached)
    moved = sum(int(not torch.equal(p.detach(), before[k])) for k, p in model.named_parameters() if p.requires_grad)
    assert moved >= 0.95 * len(before), (moved, len(before))
    assert all(p.grad is None for p in model.text_encoder.text_backbone.parameters())       # frozen (every launch script)
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    # pre-clip gradient norm of the fp16 step (unscaled by the scaler) against the fp32 product's: fp16 storage + atomics
    assert abs(float(norm) - float(norm32)) <= 0.1 * float(norm32), (float(norm), float(norm32))
    assert fallbacks.snapshot() == {}
    print(f"config-5 tiny: loss fp16 {loss_value:.4f} / fp32 {total32.item():.4f} / oracle {ref_total.item():.4f}; "
          f"grad norm fp16 {float(norm):.3f} / fp32 {float(norm32):.3f}")
