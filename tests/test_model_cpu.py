"""Host logic of the product model on CPU against the reference's golden vectors.

The product's MSDeformAttn op has no CPU path (it raises); for these host-logic tests ONLY, the op is replaced by a
test double (the oracle's C restatement, itself pinned to the reference in test_oracle_msda.py).  The GPU suite
(test_model_gpu.py) runs the same checks with the real HIP op.
"""
import pytest
import torch

import model_checks


@pytest.fixture()
def msda_double(monkeypatch):
    from oracle.msda import MSDAOracleFunction
    import ocpg_amd.models.ops.modules.ms_deform_attn as mod
    monkeypatch.setattr(mod, "MSDeformAttnFunction", MSDAOracleFunction)


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_train_step_matches_reference(golden, msda_double, tag):
    res = model_checks.run_train_step(golden("e2e_tiny"), tag, torch.device("cpu"), rtol=2e-4, atol=2e-5)
    print(res)


@pytest.mark.parametrize("fixture,tag", [("e2e_d32", "pad"), ("e2e_cfg1", "nopad")])
def test_train_step_matches_reference_other_configs(golden, msda_double, fixture, tag):
    """head_dim 32 and BASELINE config #1 (one frame, 3 levels, 1 query) through the host logic (CPU, oracle MSDeformAttn)."""
    model_checks.run_train_step(golden(fixture), tag, torch.device("cpu"), rtol=2e-4, atol=2e-5)


def test_reference_train_iteration_and_checkpoint(golden, msda_double):
    """Rows f1 / f2 pinned to the REFERENCE: engine.train_step reproduces one iteration of the reference's own
    engine.train_one_epoch (loss, pre-clip gradient norm, post-step parameters under main.py:76-99's optimizer) and
    util.checkpoint loads a checkpoint file written by the reference's util.misc.save_on_master."""
    model_checks.check_reference_iteration(golden("train_step"), torch.device("cpu"))
    model_checks.check_reference_checkpoint(golden("ckpt_ref_manifest"), torch.device("cpu"))


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_eval_tail_matches_reference(golden, msda_double, tag):
    model_checks.run_eval(golden("e2e_tiny"), tag, torch.device("cpu"), rtol=2e-4, atol=2e-5)


def test_checkpoint_wire_format(tmp_path):
    """util/checkpoint.py: the reference's checkpoint dict round-trips (legacy serialization, strict=False resume that keeps the
    current learning rates and drops gamma / milestones), and the fine-tuning filter drops exactly the class heads."""
    import argparse
    import pickle
    import cases
    from ocpg_amd.models import build_model
    from ocpg_amd.util import checkpoint as ck
    args = cases.default_args(device="cpu", **cases.TINY)
    torch.manual_seed(0)
    model, _, _ = build_model(args)
    opt = torch.optim.AdamW([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 1e-4}], weight_decay=5e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [3, 5])
    for p in model.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p) * 1e-3
    opt.step(), sched.step()
    path = tmp_path / "checkpoint0000.pth"
    ck.save_checkpoint(path, model, opt, sched, 0, argparse.Namespace(**vars(args)), grad_scaler=torch.amp.GradScaler("cpu", enabled=False))
    with open(path, "rb") as f:
        assert f.read(2) != b"PK"                      # legacy (non-zip) container, as the reference writes it
    state = torch.load(path, map_location="cpu", weights_only=False)
    assert set(state) == {"model", "optimizer", "lr_scheduler", "epoch", "args", "grad_scaler", "ocpg_rng"}
    torch.manual_seed(1)
    model2, _, _ = build_model(args)
    opt2 = torch.optim.AdamW([{"params": [p for p in model2.parameters() if p.requires_grad], "lr": 5e-5}], weight_decay=5e-4)
    sched2 = torch.optim.lr_scheduler.MultiStepLR(opt2, [2])
    state["model"]["backbone.0.body.total_ops"] = torch.zeros(1)       # profiler residue some released checkpoints carry
    missing, unexpected, epoch = ck.load_checkpoint(state, model2, opt2, sched2)
    assert not missing and not unexpected and epoch == 0
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == 5e-5                                       # the current run's learning rate wins
    assert list(sched2.milestones) == [2] and sched2.last_epoch == sched.last_epoch   # milestones of THIS run, progress restored
    kept = ck.pre_trained_model_to_finetune(state, args)
    dropped = set(state["model"]) - set(kept)
    assert dropped == {f"class_embed.{l}.{n}" for l in range(args.dec_layers) for n in ("weight", "bias")}


def test_engine_train_step(golden, msda_double):
    """engine.train_step == the reference iteration: total loss of the golden fixture, clipped gradient norm <= max_norm, the
    parameters move, and the NaN-term substitution (engine.py:53-59) yields a graph-carrying zero."""
    import cases
    from ocpg_amd import engine
    from ocpg_amd.util.misc import NestedTensor
    g = golden("e2e_tiny")
    meta = g.meta
    args, model, crit = model_checks.build_product(meta, torch.device("cpu"))
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta["nopad_sizes"], "cpu")
    model.train(), crit.train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    loss, loss_dict, norm = engine.train_step(model, crit, NestedTensor(x, mask), model_checks.text_for(B, "cpu"), targets, opt, max_norm=0.1)
    assert abs(loss - g["nopad_total"].item()) <= 2e-3 * abs(g["nopad_total"].item())
    assert float(norm) > 0.1            # the returned value is the norm BEFORE clipping (clip_grad_norm_'s contract)
    clipped = engine.total_grad_norm([p for p in model.parameters() if p.requires_grad])
    assert float(clipped) <= 0.1 * (1 + 1e-4)
    moved = sum(int(not torch.equal(v, before[k])) for k, v in model.state_dict().items())
    assert moved > 100
    a = torch.tensor(1.5, requires_grad=True)
    fixed = engine.substitute_nan_terms({"x": a * 2, "y": a * float("nan")}, [False, True])
    assert float(fixed["y"].detach()) == 0.0 and fixed["y"].requires_grad and float(fixed["x"].detach()) == 3.0
    same = {"x": a * 2}
    assert engine.substitute_nan_terms(same, [False]) is same


def test_engine_guards(golden, msda_double):
    """The two guards of the iteration: a NaN loss TERM is replaced by a graph-carrying zero and the step still runs
    (engine.py:53-59); a non-finite TOTAL raises before backward (engine.py:92-95); a degenerate target box reaches the
    reference's box assertion (util/box_ops.py:75-76) instead of silently entering the GIoU loss."""
    import cases
    from ocpg_amd import engine
    from ocpg_amd.util.misc import NestedTensor
    g = golden("e2e_tiny")
    meta = g.meta
    args, model, crit = model_checks.build_product(meta, torch.device("cpu"))
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta["nopad_sizes"], "cpu")
    model.train(), crit.train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0)

    class OneNaN(torch.nn.Module):
        def __init__(self, inner, key, everything=False):
            super().__init__()
            self.inner, self.key, self.everything, self.weight_dict = inner, key, everything, inner.weight_dict

        def forward(self, out, tg):
            losses, *rest = self.inner(out, tg)
            losses = {k: (v * float("nan") if (self.everything or k == self.key) else v) for k, v in losses.items()}
            return (losses, *rest)
    key = next(k for k in crit.weight_dict if k.startswith("loss_"))
    text = model_checks.text_for(B, "cpu")
    loss, checked, _ = engine.train_step(model, OneNaN(crit, key), NestedTensor(x.clone(), mask.clone()), text, targets, opt)
    assert loss == loss and float(checked[key].detach()) == 0.0          # finite total, the NaN term contributes zero
    with pytest.raises(FloatingPointError):
        engine.train_step(model, OneNaN(crit, key, everything=True), NestedTensor(x.clone(), mask.clone()), text, targets, opt)
    bad = [dict(t) for t in targets]
    bad[0]["boxes"] = bad[0]["boxes"].clone()
    bad[0]["boxes"][:, 2] = -0.5                                           # negative width: x1 < x0
    with pytest.raises(AssertionError, match="error boxes|boxes"):
        engine.train_step(model, crit, NestedTensor(x.clone(), mask.clone()), text, bad, opt)


def test_resume_past_a_milestone_keeps_the_decayed_lr():
    """main.py:165-180: after a resume the LR must be base * gamma ** (milestones passed), not the fresh optimizer's base LR."""
    import cases
    from ocpg_amd.models import build_model
    from ocpg_amd.util import checkpoint as ck
    import argparse
    args = cases.default_args(device="cpu", **cases.TINY)
    model, _, _ = build_model(args)

    def make():
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW([{"params": params[:10], "lr": 1e-4}, {"params": params[10:], "lr": 5e-5}], weight_decay=5e-4)
        return opt, torch.optim.lr_scheduler.MultiStepLR(opt, [2, 4])
    opt, sched = make()
    want = []
    for epoch in range(6):
        want.append([g["lr"] for g in opt.param_groups])
        opt.step()
        sched.step()
        if epoch == 2:                                   # checkpoint written at the end of epoch 2: one milestone passed
            state = {"model": model.state_dict(), "optimizer": opt.state_dict(), "lr_scheduler": sched.state_dict(), "epoch": epoch,
                     "args": argparse.Namespace(), "grad_scaler": {"scale": 1024.0, "growth_factor": 2.0, "backoff_factor": 0.5,
                                                                   "growth_interval": 2000, "_growth_tracker": 7}}
    opt2, sched2 = make()
    scaler = torch.amp.GradScaler("cpu", enabled=True)
    _, _, epoch = ck.load_checkpoint(state, model, opt2, sched2, grad_scaler=scaler)
    assert epoch == 2
    got = []
    for _ in range(epoch + 1, 6):
        got.append([g["lr"] for g in opt2.param_groups])
        opt2.step()
        sched2.step()
    for a, b in zip(got, want[epoch + 1:]):
        assert all(abs(x - y) <= 1e-12 for x, y in zip(a, b)), (got, want)
    assert sched2.get_last_lr() == [g["lr"] for g in opt2.param_groups]
    assert scaler.state_dict()["_growth_tracker"] == 7


def test_video_inference_loop(golden, msda_double):
    """inference.segment_video: clip chopping + best-query selection + un-pad / resize / sigmoid (inference_davis.py:196-250);
    one clip covering the video == the model's own eval output post-processed by hand; merge_objects == the argmax rule."""
    import torch.nn.functional as F
    from ocpg_amd import inference
    meta = golden("e2e_tiny").meta
    args, model, _ = model_checks.build_product(meta, torch.device("cpu"), dataset_file="davis")
    T, H, W = 3, meta["H"], meta["W"]
    torch.manual_seed(0)
    frames = torch.randn(T, 3, H, W)
    text = model_checks.text_for(1, "cpu")
    logits, masks = inference.segment_video(model, frames, text, clip_len=36, origin_size=(50, 70))
    assert logits.shape[0] == T and masks.shape == (T, 50, 70) and float(masks.min()) >= 0 and float(masks.max()) <= 1
    model.eval()
    with torch.no_grad():
        out = model([frames], text, [{"size": torch.tensor([H, W])}])
    best = out["pred_logits"][0].sigmoid().mean(0).max(-1)[0].argmax()
    want = F.interpolate(out["pred_masks"][0][:, best][None][:, :, :H, :W], size=(50, 70), mode="bilinear", align_corners=False).sigmoid()[0]
    assert torch.allclose(masks, want, atol=1e-6)
    l2, m2 = inference.segment_video(model, frames, text, clip_len=2)              # two clips: 2 + 1 frames
    assert m2.shape == (T, H, W) and l2.shape[0] == T
    objs = torch.stack([masks, 1 - masks])
    lab = inference.merge_objects(objs)
    assert lab.dtype == torch.uint8 and lab.shape == (T, 50, 70)
    ref = torch.cat([torch.full((1, T, 50, 70), 0.1), torch.where(objs < 0.3, torch.zeros_like(objs), objs)], 0).argmax(0)
    assert torch.equal(lab.long(), ref)


def test_collate_matches_reference(golden):
    """Row f4: collate_fn / nested_tensor_from_* == the reference's own functions on ragged clips (bit-exact)."""
    model_checks.check_collate(golden("infer_collate"))


def test_inference_loop_matches_reference(golden, msda_double):
    """Row f3: the video inference loop == the reference's own inference_davis.py statements (executed at fixture time)."""
    model_checks.check_inference_loop(golden("infer_collate"), torch.device("cpu"))
