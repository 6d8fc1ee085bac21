"""Training-step semantics of the reference driver (reference engine.py:46-118).

`train_step` is one iteration of `train_one_epoch`: autocast forward + criterion, the reference's NaN-term substitution
(engine.py:53-59: a NaN loss term is replaced by `x - x` of the first finite term, i.e. a zero that keeps the graph), the weighted
total, the non-finite guard, zero_grad, then either GradScaler scale/unscale_/clip/step/update (AMP) or backward/clip/step.
Logging (MetricLogger, image dumps, TensorBoard) is out of scope; the loss dict is returned instead.  Differences, on purpose:
the finiteness check is done on the device and read with ONE `.item()` per step (the reference's `reduce_dict` + `.item()`),
and the weighted total goes through `criterion.weighted_sum` (one reduction) when the criterion offers it.
"""
import math

import torch


def _weighted_total(criterion, loss_dict):
    if hasattr(criterion, "weighted_sum"):
        return criterion.weighted_sum(loss_dict)
    wd = criterion.weight_dict
    return sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)


def substitute_nan_terms(loss_dict):
    """engine.py:53-59, without a host sync per term: NaN terms become (finite - finite) of the first finite term."""
    keys = list(loss_dict)
    if not keys:
        return loss_dict
    stacked = torch.stack([loss_dict[k].reshape(()) for k in keys])
    bad = torch.isnan(stacked)
    if not bool(bad.any()):          # one sync; the reference syncs once per term
        return loss_dict
    good = [k for k, b in zip(keys, bad.tolist()) if not b]
    out = dict(loss_dict)
    if good:
        zero = loss_dict[good[0]] - loss_dict[good[0]]
        for k, b in zip(keys, bad.tolist()):
            if b:
                print("loss {} is Nan!!!!".format(k))
                out[k] = zero
    return out


def train_step(model, criterion, samples, captions, targets, optimizer, max_norm=0.0, amp_dtype=None, grad_scaler=None):
    """One optimisation step.  Returns (total loss as float, loss_dict, grad_total_norm).  Raises FloatingPointError on a
    non-finite total (the reference prints and sys.exit(1)s)."""
    device_type = samples.tensors.device.type
    with torch.autocast(device_type=device_type, dtype=amp_dtype, enabled=amp_dtype is not None):
        outputs = model(samples, captions, targets)
        loss_dict, *_ = criterion(outputs, targets)
        checked = substitute_nan_terms(loss_dict)
        losses = _weighted_total(criterion, checked) if checked is loss_dict else _weighted_total(_NoFastPath(criterion), checked)
    loss_value = float(losses.detach())
    if not math.isfinite(loss_value):
        raise FloatingPointError("Loss is {}, stopping training: {}".format(loss_value, {k: float(v) for k, v in loss_dict.items()}))
    optimizer.zero_grad()
    params = [p for p in model.parameters() if p.requires_grad]
    if grad_scaler is not None:
        grad_scaler.scale(losses).backward()
        if max_norm > 0:
            grad_scaler.unscale_(optimizer)
            norm = torch.nn.utils.clip_grad_norm_(params, max_norm, error_if_nonfinite=False)
        else:
            norm = total_grad_norm(params)
        grad_scaler.step(optimizer)
        grad_scaler.update()
    else:
        losses.backward()
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm, error_if_nonfinite=False) if max_norm > 0 else total_grad_norm(params)
        optimizer.step()
    return loss_value, checked, norm


class _NoFastPath:
    """Forces the generic weighted sum when the loss dict was edited (the criterion's cached table no longer matches it)."""

    def __init__(self, criterion):
        self.weight_dict = criterion.weight_dict


def total_grad_norm(parameters, norm_type=2.0):
    """util/misc.get_total_grad_norm (:583-589)."""
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, norm_type) for g in grads]), norm_type)


def train_one_epoch(model, criterion, data_loader, optimizer, device, max_norm=0.0, amp_dtype=None, grad_scaler=None, lr_scheduler=None):
    """engine.py:34-118 without the logging: iterate (samples, targets) batches; captions come from the targets."""
    model.train()
    criterion.train()
    history = []
    for samples, targets in data_loader:
        samples = samples.to(device)
        captions = [t["caption"] for t in targets]
        targets = [{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
        loss, _, norm = train_step(model, criterion, samples, captions, targets, optimizer, max_norm, amp_dtype, grad_scaler)
        history.append((loss, float(norm)))
    return history
