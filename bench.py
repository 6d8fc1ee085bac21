#!/usr/bin/env python3
"""bench.py -- clips/s of OCPG's per-clip training step on MI355X (BASELINE.json config #2 / #3).

    python bench.py [--gpus N] [--steps K] [--warmup W]         (N > 1: launched by torch.distributed.run)

One "step" = forward + criterion + backward + grad-clip + AdamW over `--clips-per-gpu` synthetic clips
(5 x 3 x 384 x 640, ResNet-101 + 4-scale deformable transformer, 5 queries, bf16 autocast, fp32 MSDeformAttn /
dynamic mask head as in the reference's --amp path).  Data-parallel over N GPUs (one process per GPU, DDP over
RCCL), per-GPU work fixed => weak scaling.  Prints ONE JSON line on rank 0.

Extra objects on that line:
  roofline     -- the hand-written kernel that dominates our HIP time (MSDeformAttn backward at the encoder shape),
                  timed live with events on the launch stream during the timed steps; achieved = algorithmic bytes
                  per launch / mean launch time, peak = 8 TB/s HBM.
  cpu_baseline -- the CPU oracle (oracle/ocpg_ref.py, a restatement of the reference's path; kind "port") timed on
                  this host for a bounded sample (one clip fwd+loss+bwd), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

T_FRAMES, HEIGHT, WIDTH = 5, 384, 640
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def model_args(device, backbone="resnet101", amp=True):
    from cases import default_args
    return default_args(device=str(device), backbone=backbone, num_frames=T_FRAMES, num_queries=5, num_feature_levels=4,
                        enc_layers=4, dec_layers=4, hidden_dim=256, dim_feedforward=2048, dropout=0.1, amp=amp,
                        text_encoder_lazy=True)


def synthetic_batch(n_clips, device, seed):
    """SURVEY.md section 8d recipe: randn clips (already 'normalised'), random text features, one box per frame."""
    from synth import synthetic_targets
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    from ocpg_amd.util.misc import NestedTensor
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n_clips, T_FRAMES, 3, HEIGHT, WIDTH, generator=g).to(device)
    mask = torch.zeros(n_clips, T_FRAMES, HEIGHT, WIDTH, dtype=torch.bool, device=device)
    text = PrecomputedText(torch.randn(n_clips, 9, 768, generator=g).to(device), torch.randn(n_clips, 768, generator=g).to(device),
                           torch.zeros(n_clips, 9, dtype=torch.bool, device=device))
    targets = synthetic_targets(n_clips, T_FRAMES, HEIGHT, WIDTH, device)
    return (lambda: NestedTensor(x.clone(), mask.clone())), text, targets


def make_optimizer(model, args):
    """AdamW with the reference's four name-based LR groups (main.py:76-99)."""
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
    groups = [g for g in groups if g["params"]]
    return torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay, fused=True)


def train_step(model, criterion, optimizer, make_samples, text, targets, args, amp_dtype):
    """engine.py:46-113 for one batch (logging all-reduce and .item() syncs left out of the hot loop)."""
    with torch.autocast(device_type="cuda", dtype=amp_dtype, enabled=amp_dtype is not None):
        out = model(make_samples(), text, targets)
        loss_dict, *_ = criterion(out, targets)
        wd = criterion.weight_dict
        loss = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if args.clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_max_norm, error_if_nonfinite=False, foreach=True)
    optimizer.step()
    return loss


def cpu_baseline(sample_frames=T_FRAMES):
    """Time the CPU oracle (a restatement of the reference path) on ONE clip: forward + criterion + backward."""
    try:
        from oracle import ocpg_ref
    except ImportError:
        return None
    return ocpg_ref.timed_baseline(model_args("cpu", amp=False), sample_frames, HEIGHT, WIDTH)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips-per-gpu", type=int, default=2)
    ap.add_argument("--backbone", default="resnet101")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"

    from ocpg_amd import _lib
    _lib.lib()          # fail loudly if the HIP extension is missing
    from ocpg_amd.models import build_model
    from ocpg_amd.models.ops.functions import ms_deform_attn_func as msda_fn

    torch.manual_seed(42 + rank)
    torch.backends.cudnn.benchmark = True
    args = model_args(device, a.backbone, amp=a.dtype != "fp32")
    model, criterion, _ = build_model(args)
    model.to(device).to(memory_format=torch.channels_last)
    criterion.to(device)
    model.train(), criterion.train()
    optimizer = make_optimizer(model, args)
    step_model = model
    if world > 1:
        step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], gradient_as_bucket_view=True,
                                                               bucket_cap_mb=64, find_unused_parameters=False)
    amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[a.dtype]
    make_samples, text, targets = synthetic_batch(a.clips_per_gpu, device, seed=42 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        train_step(step_model, criterion, optimizer, make_samples, text, targets, args, amp_dtype)
    sync()
    if not a.no_kernel_timing:
        msda_fn.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = train_step(step_model, criterion, optimizer, make_samples, text, targets, args, amp_dtype)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    kt = msda_fn.collect_kernel_timing() if not a.no_kernel_timing else {}
    assert torch.isfinite(loss).item(), "non-finite loss"

    clips = a.steps * a.clips_per_gpu * world
    line = {
        "metric": "clips/sec fwd+bwd (5x384x640, R101)", "value": clips / dt, "unit": "clips/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{a.backbone} + 4-scale deformable transformer (4 enc + 4 dec), {T_FRAMES}x{HEIGHT}x{WIDTH} clips, "
                               f"5 queries, {a.clips_per_gpu} clips/GPU/step, step = fwd + criterion + bwd + clip + AdamW",
                   "global_batch": a.clips_per_gpu * world, "parallelism": f"dp{world}", "weights": "random init",
                   "text": "random features [B,9,768] (RoBERTa bypassed, BASELINE configs #1-#4)"},
        "final_loss": float(loss),
    }
    if rank == 0:
        key = "bwd_enc"
        if key in kt and kt[key]["n"]:
            n_frames = a.clips_per_gpu * T_FRAMES
            S, M, D, LP = 5100, 8, 32, 16
            fwd_b = 4 * n_frames * (S * M * D + S * M * D + 3 * S * M * LP)
            bwd_b = fwd_b + 4 * n_frames * (S * M * D) + 4 * n_frames * (S * M * D + 3 * S * M * LP)
            us = kt[key]["ms"] / kt[key]["n"] * 1e3
            ach = bwd_b / us / 1e3
            line["roofline"] = {"bound": "hbm", "kernel": "msda_bwd (encoder shape, N=%d frames)" % n_frames, "achieved": ach,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                "launch_us": us, "algorithmic_bytes": bwd_b, "launches_timed": kt[key]["n"]}
            line["kernel_us"] = {k: v["ms"] / max(v["n"], 1) * 1e3 for k, v in kt.items()}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
