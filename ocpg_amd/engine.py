"""Training-step semantics of the reference driver (reference engine.py:46-118).

`train_step` is one iteration of `train_one_epoch`: autocast forward + criterion, the reference's NaN-term substitution
(engine.py:53-59: a NaN loss term is replaced by `x - x` of the first finite term, i.e. a zero that keeps the graph), the weighted
total, the non-finite guard, zero_grad, then either GradScaler scale/unscale_/clip/step/update (AMP) or backward/clip/step.
Logging (MetricLogger, image dumps, TensorBoard) is out of scope; the loss dict is returned instead.  Differences, on purpose:
  * everything the host needs per step -- the NaN flags of the loss terms, the weighted total and the malformed-box counter of
    the matcher / GIoU path (the reference's asserts in util/box_ops.py:75-76) -- comes back in ONE stacked readback;
  * with a process group, that vector is all-reduced first (the reference's `reduce_dict`, engine.py:79), so every rank sees
    the same total and the same verdict: a non-finite loss on one rank stops ALL ranks before anyone enters backward
    (a rank-local raise would leave the others hanging in DDP's gradient all-reduce until the RCCL timeout);
  * the weighted total goes through `criterion.weighted_sum` (one reduction) when the criterion offers it.
"""
import math

import torch
import torch.distributed as dist


def _weighted_total(criterion, loss_dict):
    if hasattr(criterion, "weighted_sum"):
        return criterion.weighted_sum(loss_dict)
    wd = criterion.weight_dict
    return sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)


def substitute_nan_terms(loss_dict, bad_flags):
    """engine.py:53-59: NaN terms become (finite - finite) of the first finite term.  `bad_flags` = per-key booleans already on
    the host (no sync here)."""
    keys = list(loss_dict)
    if not any(bad_flags):
        return loss_dict
    good = [k for k, b in zip(keys, bad_flags) if not b]
    out = dict(loss_dict)
    if good:
        zero = loss_dict[good[0]] - loss_dict[good[0]]
        for k, b in zip(keys, bad_flags):
            if b:
                print("loss {} is Nan!!!!".format(k))
                out[k] = zero
    return out


def _box_error_flag(device):
    from .models import matcher
    flag = matcher._BOX_ERRORS.get(device)
    return flag


def train_step(model, criterion, samples, captions, targets, optimizer, max_norm=0.0, amp_dtype=None, grad_scaler=None):
    """One optimisation step.  Returns (total loss as float -- averaged over ranks like engine.py:79-86, loss_dict,
    grad_total_norm).  Raises FloatingPointError on a non-finite total (the reference prints and sys.exit(1)s) and
    AssertionError on malformed boxes (util/box_ops.py:75-76) -- on every rank together."""
    device = samples.tensors.device
    with torch.autocast(device_type=device.type, dtype=amp_dtype, enabled=amp_dtype is not None):
        outputs = model(samples, captions, targets)
        loss_dict, *_ = criterion(outputs, targets)
        keys = list(loss_dict)
        total = _weighted_total(criterion, loss_dict)
        # one vector for the host: [per-term NaN flags..., weighted total, malformed-box count]
        terms = torch.stack([loss_dict[k].detach().reshape(()).float() for k in keys]) if keys else torch.zeros(0, device=device)
        flag = _box_error_flag(device)
        report = torch.cat([torch.isnan(terms).float(), total.detach().reshape(1).float(),
                            (flag.reshape(1).float() if flag is not None else torch.zeros(1, device=device))])
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
            dist.all_reduce(report)               # NaN in any rank's total stays NaN; flags / counters add up
        host = report.tolist()                    # THE sync of the step
        bad = [x > 0 for x in host[:len(keys)]]
        loss_value = host[len(keys)] / world
        n_bad_boxes = int(host[len(keys) + 1])
        if n_bad_boxes:
            if flag is not None:
                flag.zero_()
            raise AssertionError(f"error boxes: {n_bad_boxes} malformed (x1 < x0 or y1 < y0, or NaN) box set(s)")
        checked = substitute_nan_terms(loss_dict, bad)
        losses = total if checked is loss_dict else _weighted_total(_NoFastPath(criterion), checked)
        if checked is not loss_dict:              # terms were replaced: the total the ranks agree on is the repaired one
            rep = losses.detach().reshape(1).float().clone()
            if world > 1:
                dist.all_reduce(rep)
            loss_value = float(rep) / world
    if not math.isfinite(loss_value):
        raise FloatingPointError("Loss is {}, stopping training: {}".format(loss_value, {k: float(v.detach()) for k, v in loss_dict.items()}))
    optimizer.zero_grad()
    params = [p for p in model.parameters() if p.requires_grad]
    fused = hasattr(optimizer, "step_clip")        # optim.ClipAdamW: (unscale +) norm + clip + AdamW in three launches, no host sync
    if fused and grad_scaler is not None:
        grad_scaler.scale(losses).backward()
        grad_scaler.step(optimizer, max_norm=max_norm)
        grad_scaler.update()
        norm = optimizer.grad_norm
    elif fused:
        losses.backward()
        norm = optimizer.step_clip(max_norm)
    elif grad_scaler is not None:
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


@torch.no_grad()
def evaluate_referred_masks(model, data_loader, postprocessor, device, amp_dtype=None):
    """The A2D-Sentences / JHMDB-Sentences evaluation loop (reference engine.py:126-194) up to the metrics that need no COCO api:
    per sample the post-processed mask of the highest-scoring query against the ground-truth mask of the annotated frame at the
    original resolution -> precision@K, overall IoU, mean IoU (ocpg_amd/metrics.py).  Targets carry `orig_size`, `size` and the
    ground truth as `gt_mask` [H0, W0] (the reference reads it from the dataset's COCO-format annotation file instead).  With a
    process group the running sums are all-reduced, so every rank returns the same numbers."""
    from . import metrics
    from .util.misc import targets_to
    model.eval()
    state = None
    for samples, targets in data_loader:
        samples = samples.to(device)
        captions = [t["caption"] for t in targets]
        targets = targets_to(targets, device)
        with torch.autocast(device_type=torch.device(device).type, dtype=amp_dtype, enabled=amp_dtype is not None):
            outputs = model(samples, captions, targets)
        orig = torch.stack([t["orig_size"] for t in targets], dim=0)
        size = torch.stack([t["size"] for t in targets], dim=0)
        for p, t in zip(postprocessor(outputs, orig, size), targets):
            best = metrics.select_best_query(p["scores"][None], p["masks"][None, :, 0])      # [1, H0, W0]
            state = metrics.accumulate(state, best, t["gt_mask"][None].bool())
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world > 1:
        # every rank enters the collectives, also one whose loader was empty (the reference always reaches its all_gather /
        # barrier, engine.py:170-176; a rank-local early return would leave the others waiting)
        dev = torch.device(device)
        ious = torch.cat(state[0]) if state is not None else torch.zeros(0, device=dev)
        gathered = [None] * world
        dist.all_gather_object(gathered, ious.cpu())
        sums = (torch.stack([state[1], state[2]]) if state is not None else torch.zeros(2, device=dev)).to(torch.float64)
        dist.all_reduce(sums)
        state = ([g.to(dev) for g in gathered], sums[0].float(), sums[1].float())
    if state is None or sum(g.numel() for g in state[0]) == 0:
        return {}
    return metrics.summarize(state)
