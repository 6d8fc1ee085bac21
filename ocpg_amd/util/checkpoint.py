"""Checkpoint wire format of the reference's training driver.

Save: the dict of main.py:226-236 -- {'model', 'optimizer', 'lr_scheduler', 'epoch', 'args', 'grad_scaler'} -- written with the
legacy (non-zip) serialization the reference uses (util/misc.py:444-446), so files are interchangeable in both directions.
Resume: main.py:150-180 -- model with strict=False (keys ending in total_params / total_ops are profiler residue and ignored),
optimizer state restored but the CURRENT base learning rates kept, 'gamma' / 'milestones' dropped from the scheduler state so
that a changed --lr_drop takes effect, then -- main.py:175-177 -- the scheduler's base_lrs are re-seeded from the groups'
initial_lr and the learning rates are recomputed in closed form for the restored epoch (`lr_scheduler.step(last_epoch)` in the
reference), so a run resumed past a milestone continues at the decayed rate; the GradScaler state is restored when both sides
have one (main.py:178-179).  Fine-tuning from Ref-COCO weights: utils.py:5-13 (class heads dropped).
"""
import bisect
import copy
import warnings

import torch


def save_checkpoint(path, model, optimizer, lr_scheduler, epoch, args, grad_scaler=None, is_main=True):
    """main.py:226-236 + save_on_master.  `model` is the un-wrapped module (model_without_ddp)."""
    if not is_main:
        return
    state = {"model": model.state_dict(), "optimizer": optimizer.state_dict(), "lr_scheduler": lr_scheduler.state_dict(), "epoch": epoch,
             "args": args}
    if grad_scaler is not None:
        state["grad_scaler"] = grad_scaler.state_dict()
    state["ocpg_rng"] = _fused_rng().get_rng_state()     # extra key (the reference's loader reads only the keys it knows)
    torch.save(state, path, _use_new_zipfile_serialization=False)


def _fused_rng():
    from ..models.ops.functions import fused_ln_func
    return fused_ln_func


def _recompute_lrs(optimizer, lr_scheduler):
    """main.py:176-177: base_lrs <- initial_lr, then the LR of the restored epoch.  MultiStepLR is chainable: only its closed
    form (what `step(epoch)` evaluates) gives base * gamma ** (milestones passed) after a reload."""
    lr_scheduler.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
    milestones, gamma = getattr(lr_scheduler, "milestones", None), getattr(lr_scheduler, "gamma", None)
    if milestones is not None and gamma is not None:
        passed = bisect.bisect_right(sorted(milestones.elements()) if hasattr(milestones, "elements") else sorted(milestones),
                                     lr_scheduler.last_epoch)
        lrs = [base * gamma ** passed for base in lr_scheduler.base_lrs]
        for g, lr in zip(optimizer.param_groups, lrs):
            g["lr"] = lr
        lr_scheduler._last_lr = lrs
    else:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lr_scheduler.step(lr_scheduler.last_epoch)


def load_checkpoint(path_or_state, model, optimizer=None, lr_scheduler=None, eval_only=False, grad_scaler=None):
    """main.py:150-180.  Returns (missing_keys, unexpected_keys, epoch or None).  Reference checkpoints pickle their argparse
    Namespace, hence weights_only=False: only load files you trust, exactly as with the reference."""
    ckpt = path_or_state if isinstance(path_or_state, dict) else torch.load(path_or_state, map_location="cpu", weights_only=False)
    missing, unexpected = model.load_state_dict(ckpt["model"], strict=False)
    unexpected = [k for k in unexpected if not (k.endswith("total_params") or k.endswith("total_ops"))]
    epoch = None
    if not eval_only and optimizer is not None and lr_scheduler is not None and all(k in ckpt for k in ("optimizer", "lr_scheduler", "epoch")):
        groups = copy.deepcopy(optimizer.param_groups)
        optimizer.load_state_dict(ckpt["optimizer"])
        for pg, old in zip(optimizer.param_groups, groups):
            pg["lr"] = old["lr"]
            if "initial_lr" in old:
                pg["initial_lr"] = old["initial_lr"]
        sched = dict(ckpt["lr_scheduler"])
        sched.pop("gamma", None)
        sched.pop("milestones", None)
        lr_scheduler.load_state_dict(sched)
        _recompute_lrs(optimizer, lr_scheduler)
        if grad_scaler is not None and "grad_scaler" in ckpt:
            grad_scaler.load_state_dict(ckpt["grad_scaler"])
        if "ocpg_rng" in ckpt:
            _fused_rng().set_rng_state(ckpt["ocpg_rng"])
        epoch = ckpt["epoch"]
    return list(missing), unexpected, epoch


def pre_trained_model_to_finetune(checkpoint, args):
    """utils.py:5-13: keep everything but the class heads (the fine-tuning dataset has a different number of classes)."""
    state = dict(checkpoint["model"])
    n = args.dec_layers + 1 if getattr(args, "two_stage", False) else args.dec_layers
    for l in range(n):
        state.pop(f"class_embed.{l}.weight", None)
        state.pop(f"class_embed.{l}.bias", None)
    return state
