#!/usr/bin/env python3
"""bench.py -- clips/s of OCPG's per-clip training step on MI355X (BASELINE.json config #2 / #3).

    python bench.py [--gpus N] [--steps K] [--warmup W]         (N > 1: launched by torch.distributed.run)

One "step" = forward + criterion + backward + grad-clip + AdamW over `--clips-per-gpu` synthetic clips
(5 x 3 x 384 x 640, ResNet-101 + 4-scale deformable transformer, 5 queries, bf16 autocast, fp32 MSDeformAttn /
dynamic mask head as in the reference's --amp path).  Data-parallel over N GPUs (one process per GPU, DDP over
RCCL), per-GPU work fixed => weak scaling.  Prints ONE JSON line on rank 0.

Extra objects on that line:
  roofline     -- the hand-written kernel that dominates our HIP time (the grad_value scatter of the MSDeformAttn backward
                  at the encoder shape), timed live with events on the launch stream during the timed steps; achieved =
                  algorithmic bytes per launch / mean launch time, peak = 8 TB/s HBM.
  kernels      -- the same for the other hand-written kernels of the step (per-launch algorithmic bytes or FLOPs, mean
                  microseconds from live events, fraction of the HBM / vector-FP32 peak), sorted by time per step.
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
import torch.utils._python_dispatch
import torch.utils._pytree

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_FRAMES, HEIGHT, WIDTH = 5, 384, 640        # BASELINE config #2/#3/#4 clip; --frames/--height/--width override (config #5: 8x480x854)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def model_args(device, backbone="resnet101", amp=True, roberta=False):
    from ocpg_amd.opts import default_args
    return default_args(device=str(device), backbone=backbone, num_frames=T_FRAMES, num_queries=5, num_feature_levels=4,
                        enc_layers=4, dec_layers=4, hidden_dim=256, dim_feedforward=2048, dropout=0.1, amp=amp,
                        text_encoder_lazy=not roberta)


def synthetic_batch(n_clips, device, seed, roberta=False):
    """SURVEY.md section 8d recipe: randn clips (already 'normalised'), random text features (or, for config #5, captions
    that go through the random-init RoBERTa-base), one box per frame.  Widths that are not a multiple of 32 are padded
    like collate_fn does (util/misc.py:302), with the padding mask set."""
    from ocpg_amd.util.synthetic import synthetic_targets
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    from ocpg_amd.util.misc import NestedTensor, tag_rect_mask
    g = torch.Generator(device="cpu").manual_seed(seed)
    hp, wp = (HEIGHT + 31) // 32 * 32, (WIDTH + 31) // 32 * 32
    x = torch.zeros(n_clips, T_FRAMES, 3, hp, wp)
    x[..., :HEIGHT, :WIDTH] = torch.randn(n_clips, T_FRAMES, 3, HEIGHT, WIDTH, generator=g)
    x = x.to(device)
    mask = torch.ones(n_clips, T_FRAMES, hp, wp, dtype=torch.bool)
    mask[..., :HEIGHT, :WIDTH] = False
    mask = mask.to(device)
    if roberta:
        words = ["the", "person", "on", "left", "riding", "a", "red", "bike", "near", "tree", "small", "dog"]
        text = [" ".join(words[(i + j) % len(words)] for j in range(10)) for i in range(n_clips)]
    else:
        text = PrecomputedText(torch.randn(n_clips, 9, 768, generator=g).to(device), torch.randn(n_clips, 768, generator=g).to(device),
                               torch.zeros(n_clips, 9, dtype=torch.bool, device=device))
    targets = synthetic_targets(n_clips, T_FRAMES, HEIGHT, WIDTH, device)
    # the collate step knows every frame's valid extent on the host (as util.misc.collate_fn does) and says so on the mask
    valid_hw = [(HEIGHT, WIDTH)] * (n_clips * T_FRAMES)
    return (lambda: NestedTensor(x.clone(), tag_rect_mask(mask.clone(), valid_hw))), text, targets


def make_optimizer(model, args, fused=True):
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
    if fused and groups and groups[0]["params"][0].is_cuda and os.environ.get("OCPG_CLIP_ADAMW", "1") != "0":
        from ocpg_amd.optim import ClipAdamW            # clip + AdamW as three HIP launches (csrc/adamw.hip); torch.optim.AdamW state layout
        return ClipAdamW(groups, lr=args.lr, weight_decay=args.weight_decay)
    return torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay, fused=fused)


def wrap_ddp(model, local_rank=None):
    """main.py:62's DistributedDataParallel, configured for this model: bucket views (no grad copy), 64-MB buckets (xGMI is
    point-to-point: fewer, larger all-reduces), no unused-parameter search (every trainable parameter receives a gradient:
    tests/test_ddp_gloo.py), no buffer broadcast (the only buffers are frozen BN statistics, identical on every rank; re-
    broadcasting them would also invalidate FrozenBatchNorm2d's cached scale/shift).  local_rank None = CPU (gloo tests)."""
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=None if local_rank is None else [local_rank],
                                                     gradient_as_bucket_view=True, bucket_cap_mb=64, find_unused_parameters=False,
                                                     broadcast_buffers=False)


def forward_backward(model, criterion, samples, text, targets, amp_dtype, num_boxes=None, keep=None, scaler=None):
    """engine.py:50-62 + backward: forward, criterion, weighted sum, backward (grads accumulate into .grad)."""
    with torch.autocast(device_type=samples.tensors.device.type, dtype=amp_dtype, enabled=amp_dtype is not None):
        out = model(samples, text, targets)
        if num_boxes is not None:
            out["num_boxes"] = num_boxes
        loss_dict, *_ = criterion(out, targets)
        loss = criterion.weighted_sum(loss_dict)        # = sum(loss_dict[k] * weight_dict[k]) (engine.py:56), one reduction
    (scaler.scale(loss) if scaler is not None else loss).backward()
    from ocpg_amd.models import amp_cache
    amp_cache.join_wgrad()      # (a no-op after FusedCast.backward; a capture must not end with an un-joined side stream)
    if keep is not None:        # static graph outputs (the usual whole-network-capture rule: keep them referenced)
        keep.update(out=out, loss_dict=loss_dict, loss=loss)
    return loss.detach()


class EagerStep:
    """engine.py:46-113 for one batch, eager launches (DDP overlaps the gradient all-reduce with backward).  With fp16 the
    reference's GradScaler path is used (scale, unscale_, clip, scaler.step, update); bf16 / fp32 need no scaler."""

    def __init__(self, model, ddp_model, criterion, optimizer, make_samples, text, targets, args, amp_dtype):
        self.__dict__.update(locals())
        self.scaler = torch.amp.GradScaler("cuda") if amp_dtype == torch.float16 else None
        self.grad_norm = None           # pre-clip total gradient norm of the last step (device scalar)

    def __call__(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = forward_backward(self.ddp_model, self.criterion, self.make_samples(), self.text, self.targets, self.amp_dtype,
                                scaler=self.scaler)
        if hasattr(self.optimizer, "step_clip"):
            if self.scaler is None:
                self.grad_norm = self.optimizer.step_clip(self.args.clip_max_norm)     # norm + clip + AdamW: three launches
            else:                                                                      # + unscale and the skip-on-overflow, on the device
                self.scaler.step(self.optimizer, max_norm=self.args.clip_max_norm)
                self.scaler.update()
                self.grad_norm = self.optimizer.grad_norm
            return loss
        if self.scaler is not None:
            self.scaler.unscale_(self.optimizer)
        if self.args.clip_max_norm > 0:
            self.grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.args.clip_max_norm, error_if_nonfinite=False,
                                                            foreach=True)
        if self.scaler is not None:
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            self.optimizer.step()
        return loss


class GraphStep:
    """Same step with forward + criterion + backward captured once into a HIP graph and replayed (launch-bound inner
    loop: ~9 500 kernels per step).  Outside the graph, per step: copy the batch into the static input buffers, the
    criterion's num_boxes all-reduce, (N>1) one flat gradient all-reduce over RCCL, grad clip, AdamW."""

    INIT_SCALE = None       # fp16: the GradScaler's initial scale (None = torch's 65536; tests compare gradients at a scale that does not overflow)

    def __init__(self, model, criterion, optimizer, make_samples, text, targets, args, amp_dtype, world):
        from ocpg_amd.util.misc import NestedTensor
        self.model, self.criterion, self.optimizer, self.args, self.world = model, criterion, optimizer, args, world
        self.make_samples, self.targets = make_samples, targets
        self.fence = os.environ.get("OCPG_GRAPH_FENCE") == "1"       # diagnostic: host syncs around every replay
        # fp16: the reference's GradScaler path (engine.py:98-104).  The scale is a device scalar, so `scale(loss).backward()` is
        # captured; unscale_ / clip / step (with its inf check) / update stay eager after the replay
        self.scaler = (torch.amp.GradScaler("cuda", **({"init_scale": self.INIT_SCALE} if self.INIT_SCALE else {}))
                       if amp_dtype == torch.float16 else None)
        # captions (config #5): tokenisation uploads host tensors, which a capture cannot contain.  The FROZEN text backbone therefore
        # runs eagerly before every replay (same work per step as the eager path, `freeze_text_encoder` as in every launch script) and
        # hands its outputs to the captured step through static buffers; the trainable resizers stay inside the graph.
        self.captions, self.amp_dtype = None, amp_dtype
        if isinstance(text, (list, tuple)) and text and isinstance(text[0], str):
            if not getattr(model.text_encoder, "freeze_text_encoder", False):
                raise RuntimeError("captured step needs a frozen text encoder")
            from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
            self.captions = text
            text = PrecomputedText(*[t.clone() for t in self._encode_text()])
        self.text = text
        first = make_samples()
        self.x, self.mask = first.tensors.clone(), first.mask.clone()
        # the collate step's host-side knowledge of every frame's valid extent (util.misc.tag_rect_mask) must survive the copies:
        # without it the captured step takes the uncached path (position encodings, level masks, 18 masked_fill passes per step)
        self.mask_key = getattr(first.mask, "_ocpg_key", None)
        self.num_boxes = criterion.global_num_boxes(targets, self.x.device).clone()
        criterion.iter_device = torch.zeros((), device=self.x.device)       # the criterion's call counter, device-resident
        self.calls_per_fwd = args.dec_layers
        self.params = [p for p in model.parameters() if p.requires_grad]
        # AccumulateGrad nodes remember the stream they were created on and are handed from one autograd graph to the next for
        # as long as ANY graph is alive (a parameter only holds a weak reference).  An earlier eager step on the default stream
        # whose graph is still reachable through a reference cycle would therefore pull the NULL stream into the capture
        # (cross-stream gradient accumulation) and hipStreamEndCapture segfaults on it.  Drop such graphs first.
        import gc
        criterion._last = None
        gc.collect()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):                      # warm-up on the capture stream: MIOpen find, FFT plans, workspaces
                optimizer.zero_grad(set_to_none=True)
                forward_backward(model, criterion, NestedTensor(self.x.clone(), self._mask()), text, targets, amp_dtype, self.num_boxes,
                                 scaler=self.scaler)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.zero_grad(set_to_none=True)
        from ocpg_amd.models.ops.functions.fused_ln_func import GraphRng
        self.rng = GraphRng(self.x.device)                      # dropout generator state of this capture (device-resident base)
        # The captured step's outputs (model outputs, the 36 losses) stay referenced for the lifetime of the graph (static
        # outputs); self.check() compares a replay against an eager step before the timed region.
        self.static = {}
        self.memset_nodes_replaced = 0
        self._capture(side, text, targets)
        self.grads = [p.grad for p in self.params]
        assert all(g is not None for g in self.grads), "a trainable parameter received no gradient"
        self.flat = None

    def _capture(self, side, text, targets):
        from ocpg_amd.util.misc import NestedTensor
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)      # instantiated below, after the memset nodes are repaired
        # capture on the warm-up stream: the library's hipBLASLt workspace is per (device, stream) and was allocated there
        # (a hipMalloc inside the capture would invalidate it)
        # N > 1: RCCL's watchdog thread may poll events while we capture; only THIS thread's unsafe calls should invalidate the capture
        with self.rng, torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local" if self.world > 1 else "global"):
            self.loss = forward_backward(self.model, self.criterion, NestedTensor(self.x.clone(), self._mask()), text, targets, self.amp_dtype,
                                         self.num_boxes, keep=self.static, scaler=self.scaler)
            self.rng.advance()          # last node of the step: the next replay draws fresh dropout masks
        self.rng.finalize()
        self._repair_and_instantiate(self.graph, stats=True)
        self.graphs = [self.graph]

    def _repair_and_instantiate(self, graph, stats=False):
        """memset nodes (torch's reduction semaphores) replay with a corrupted pattern on this ROCm: swap them for kernel nodes."""
        import ctypes
        from ocpg_amd import _lib
        n_fixed = ctypes.c_int(0)
        _lib.check(_lib.lib().ocpg_graph_replace_memsets(graph.raw_cuda_graph(), ctypes.byref(n_fixed)), "ocpg_graph_replace_memsets")
        self.memset_nodes_replaced += n_fixed.value
        if stats:
            st = (ctypes.c_longlong * 9)()
            _lib.check(_lib.lib().ocpg_graph_stats(graph.raw_cuda_graph(), st), "ocpg_graph_stats")
            self.graph_stats = dict(zip(("nodes", "edges", "roots", "max_out_degree", "max_in_degree", "kernel_nodes", "memset_nodes",
                                         "memcpy_nodes", "other_nodes"), [int(v) for v in st]))
            cap = 4096
            rows = (ctypes.c_longlong * (4 * cap))()
            n_cp = _lib.lib().ocpg_graph_memcpy_nodes(graph.raw_cuda_graph(), rows, cap)
            self.memcpy_nodes = [tuple(rows[4 * i:4 * i + 4]) for i in range(max(0, min(n_cp, cap)))]     # (kind, bytes, src, dst)
        graph.instantiate()

    def replay(self):
        for g in self.graphs:
            g.replay()
        self.rng.replayed()

    def replay_and_reduce(self):
        """One step's forward + criterion + backward, gradients averaged over the ranks.  Single graph: ONE flat all-reduce after the
        replay (not overlapped; SegmentedGraphStep overlaps)."""
        self.replay()
        if self.world > 1:
            flat = torch._utils._flatten_dense_tensors(self.grads)
            dist.all_reduce(flat)
            flat.div_(self.world)
            torch._foreach_copy_(self.grads, list(torch._utils._unflatten_dense_tensors(flat, self.grads)))

    def _encode_text(self):
        dev = next(self.model.parameters()).device
        with torch.no_grad(), torch.autocast(device_type=dev.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            return self.model.text_encoder(self.captions, dev)

    def _mask(self):
        m = self.mask.clone()
        if self.mask_key is not None:
            m._ocpg_key = self.mask_key
        return m

    def check(self, eager_loss=None, rtol=None):
        """The FIRST replay against an eager step at the same parameters ON THE SAME DROPOUT STREAMS: a replay r of a graph captured at
        dropout counter c0 draws what eager step r would draw (GraphRng; torch's own generator registers with the graph the same way), so
        with both generator states put back after the eager step the two losses agree to the arithmetic noise that
        tests/test_graph_gpu.py::test_whole_step_graph_matches_eager establishes (2e-3 relative under bf16 / fp16 autocast: MIOpen's
        solvers are not run-to-run reproducible; 1e-5 in fp32) -- not merely "the same ballpark".  eager_loss (an eager step on OTHER
        masks, from the caller) only has to be in the ballpark.  Gradients finite."""
        from ocpg_amd.models.ops.functions import fused_ln_func
        from ocpg_amd.util.misc import NestedTensor
        dev = self.x.device
        hip0, cuda0 = fused_ln_func.get_rng_state(), torch.cuda.get_rng_state(dev)
        # (the eager backward ACCUMULATES into the graphs' static gradient tensors; the replay below rewrites them)
        ref = float(forward_backward(self.model, self.criterion, NestedTensor(self.x.clone(), self._mask()), self.text, self.targets, self.amp_dtype,
                                     self.num_boxes, scaler=self.scaler))
        torch.cuda.synchronize()
        fused_ln_func.set_rng_state(hip0)
        torch.cuda.set_rng_state(cuda0, dev)
        self.replay()
        torch.cuda.synchronize()
        loss = float(self.loss)
        tol = rtol if rtol is not None else (1e-5 if self.amp_dtype is None else 2e-3)
        # (fp16: a replay at the initial loss scale may overflow -- that is the scaler's business, not a capture failure)
        ok = loss == loss and abs(loss - ref) <= tol * abs(ref) and (self.scaler is not None or all(bool(torch.isfinite(g).all()) for g in self.grads))
        if ok and eager_loss is not None:
            ok = abs(loss - eager_loss) <= 0.25 * abs(eager_loss)
        self.check_result = {"replay_loss": loss, "eager_same_masks": ref, "rel_diff": abs(loss - ref) / max(abs(ref), 1e-30), "tolerance": tol}
        if not ok:
            raise RuntimeError(f"graph replay disagrees with eager on the same dropout masks: {loss} vs {ref} (tolerance {tol}); other masks: {eager_loss}")

    def _phase(self, name):
        """OCPG_STEP_PHASES=1 (diagnostic): host wall time per phase of a step with a device sync at every boundary -> self.phases."""
        if not self.__dict__.setdefault("_ph_on", os.environ.get("OCPG_STEP_PHASES") == "1"):
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        ph = self.__dict__.setdefault("phases", {})
        if self.__dict__.get("_ph_last") is not None and name is not None:
            ph[name] = ph.get(name, 0.0) + now - self._ph_last
        self._ph_last = now

    def __call__(self):
        self._phase(None)
        s = self.make_samples()
        assert getattr(s.mask, "_ocpg_key", None) == self.mask_key, "the captured step assumes the valid extents it was captured with"
        self.x.copy_(s.tensors), self.mask.copy_(s.mask)
        self.num_boxes.copy_(self.criterion.global_num_boxes(self.targets, self.x.device))
        self._phase("inputs")
        if self.captions is not None:
            for dst, src in zip(self.text, self._encode_text()):
                dst.copy_(src)
        self._phase("text encoder (eager)")
        if self.fence:
            torch.cuda.synchronize()
        self.replay_and_reduce()
        self._phase("graph replay")
        if self.fence:
            torch.cuda.synchronize()
        self.criterion.iter_device += self.calls_per_fwd
        if hasattr(self.optimizer, "step_clip"):
            if self.scaler is None:
                self.grad_norm = self.optimizer.step_clip(self.args.clip_max_norm)     # norm + clip + AdamW: three launches
            else:                                                                      # + unscale and the skip-on-overflow, on the device
                self.scaler.step(self.optimizer, max_norm=self.args.clip_max_norm)
                self.scaler.update()
                self.grad_norm = self.optimizer.grad_norm
            self._phase("clip + step")
            return self.loss
        if self.scaler is not None:
            self.scaler.unscale_(self.optimizer)
        if self.args.clip_max_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.params, self.args.clip_max_norm, error_if_nonfinite=False, foreach=True)
        if self.scaler is not None:
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            self.optimizer.step()
        self._phase("unscale + clip + step + update")
        return self.loss


class SegmentedGraphStep(GraphStep):
    """The captured step cut into THREE graphs at activations of the backbone's backward, so that the gradient all-reduce of one part
    overlaps the backward of the next (main.py:62's DistributedDataParallel overlaps bucket by bucket; a single captured graph
    cannot: its gradients only exist when the replay ends):
        G1  forward + criterion + backward of everything behind the backbone      -> all-reduce bucket 0 (neck, transformer, heads)
        G2  backward of the late backbone                                          -> all-reduce bucket 1   | while G2 / G3 run
        G3  backward of the early backbone                                         -> all-reduce bucket 2   (the only exposed one)
    ResNet (configs #1-#3): G2 = layer4 + the later half of layer3, G3 = the first half of layer3 + layer2.  Video-Swin (configs #4 / #5):
    G2 = stages 3 and 2 with the patch mergings in front of them, G3 = stages 1 and 0 + the patch embedding (stage outputs feed the
    neck AND the next stage, as the ResNet layers do).  Each part's autocast parameters have their OWN fused-cast node
    (amp_cache.set_groups), so their fp32 gradients are born as views of ONE flat buffer per part: the buckets are those buffers
    themselves (no flatten / copy-back), plus one small flattened remainder for parameters outside the cast (fp32 islands).  fp16: the
    SCALED loss is differentiated (engine.py:98-104); the buckets carry scaled gradients and ClipAdamW unscales after the waits."""

    SPLIT = 10          # layer3 blocks [0, SPLIT) belong to G3, [SPLIT, 23) to G2 (ResNet-101: ~12 M / ~29 M backbone parameters)

    @staticmethod
    def _kind(model):
        body = getattr(model.backbone[0], "body", None)
        if body is None:
            return None
        if all(hasattr(body, f"layer{i}") for i in (2, 3, 4)) and len(body.layer3) >= 2:
            return "resnet"
        if hasattr(body, "layers") and hasattr(body, "downsamples") and hasattr(body, "patch_embed") and len(body.layers) in (3, 4):
            return "swin"
        return None

    @staticmethod
    def supported(model, amp_dtype):
        return amp_dtype in (torch.bfloat16, torch.float16) and SegmentedGraphStep._kind(model) is not None

    @staticmethod
    def plan(model):
        """-> (group_of, cuts, hops): group_of(parameter name) in 0..4 (0 = behind the backbone), cuts = [(key, module)] in forward order
        (the module's output becomes a detached leaf in the captured forward), hops = per later graph [(from key, group, to key | None)]:
        `grad(orig[from], P[group] + [leaf[to]], grad_outputs = what has arrived at leaf[from])`."""
        body = model.backbone[0].body
        if SegmentedGraphStep._kind(model) == "resnet":
            k = min(SegmentedGraphStep.SPLIT, len(body.layer3) - 1)

            def group_of(n):     # 0 behind the backbone | 1 layer4 | 2 layer3[k:] | 3 layer3[:k] | 4 layer2 (and the frozen stem / layer1)
                if not n.startswith("backbone.0.body."):
                    return 0
                q = n.split(".")
                if q[3] == "layer4":
                    return 1
                if q[3] == "layer3":
                    return 2 if int(q[4]) >= k else 3
                return 4
            cuts = (("f8", body.layer2), ("mid", body.layer3[k - 1]), ("f16", body.layer3), ("f32", body.layer4))
            hops = ([("f32", 1, "f16"), ("f16", 2, "mid")], [("mid", 3, "f8"), ("f8", 4, None)])
            return group_of, cuts, hops
        n_st = len(body.layers)          # 4 (3 when the last stage is not built): stage i owns the patch merging in front of it
        # (downsamples[i - 1]), stage 0 the patch embedding; the two LAST stages go to G2, the earlier ones to G3
        group = {n_st - 1: 1, n_st - 2: 2, n_st - 3: 3}
        if n_st == 4:
            group[0] = 4

        def group_of(n):     # 0 behind the backbone | 1 last stage | 2 the stage before it | 3, 4 the stages before that
            if not n.startswith("backbone.0.body."):
                return 0
            q = n.split(".")
            st = int(q[4]) if q[3] == "layers" else int(q[4]) + 1 if q[3] == "downsamples" else 0
            return group[st]
        cuts = tuple((f"s{i}", body.layers[i]) for i in range(n_st))
        hops = ([(f"s{n_st - 1}", 1, f"s{n_st - 2}"), (f"s{n_st - 2}", 2, f"s{n_st - 3}")],
                [(f"s{i}", group[i], f"s{i - 1}" if i > 0 else None) for i in range(n_st - 3, -1, -1)])
        return group_of, cuts, hops

    def __init__(self, model, criterion, optimizer, make_samples, text, targets, args, amp_dtype, world):
        from ocpg_amd.models import amp_cache
        self.kind = self._kind(model)
        self.group_of, self.cut_modules, self.hops = self.plan(model)
        amp_cache.set_groups(model, self.group_of)       # one fused-cast node (one flat gradient buffer) per part: see _capture
        super().__init__(model, criterion, optimizer, make_samples, text, targets, args, amp_dtype, world)

    def _capture(self, side, text, targets):
        from ocpg_amd.util.misc import NestedTensor
        model, criterion = self.model, self.criterion
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        P = [[p for n, p in named if self.group_of(n) == i] for i in range(5)]
        mode = "thread_local" if self.world > 1 else "global"
        self.graphs = [torch.cuda.CUDAGraph(keep_graph=True) for _ in range(3)]
        st = self.static
        pool = []

        def part(i):        # the capture of graph i (graphs 1, 2 share graph 0's pool: gradients handed from one to the next)
            if i == 0:
                return torch.cuda.graph(self.graphs[0], stream=side, capture_error_mode=mode)
            if not pool:
                pool.append(self.graphs[0].pool())
            return torch.cuda.graph(self.graphs[i], stream=side, pool=pool[0], capture_error_mode=mode)

        def forward_loss():
            with torch.autocast(device_type=self.x.device.type, dtype=self.amp_dtype):
                out = model(NestedTensor(self.x.clone(), self._mask()), text, targets)
                out["num_boxes"] = self.num_boxes
                loss_dict, *_ = criterion(out, targets)
                loss = criterion.weighted_sum(loss_dict)
            st.update(out=out, loss_dict=loss_dict, loss=loss)
            return self.scaler.scale(loss) if self.scaler is not None else loss
        with self.rng:
            glists, groups, keep = self.segmented_backward(forward_loss, P, self.cut_modules, self.hops, part, last=self.rng.advance)
        st.update(keep)
        st["g0"], st["g1"], st["g2"] = glists
        loss = st["loss"]
        seg_params = [[p for k in gr for p in P[k]] for gr in groups]
        assert sum(len(x_) for x_ in seg_params) == len(named), "a trainable parameter belongs to no segment"
        self.rng.finalize()
        for i, gr in enumerate(self.graphs):
            self._repair_and_instantiate(gr, stats=(i == 0))
        self.loss = loss.detach()
        for ps, gs in zip(seg_params, (st["g0"], st["g1"], st["g2"])):
            for p_, g_ in zip(ps, gs):
                assert g_ is not None, "a trainable parameter received no gradient"
                p_.grad = g_            # static tensors of the graphs' pool: every replay refills them
        self.buckets = self.build_buckets((st["g0"], st["g1"], st["g2"]), self.x.device)
        self.bucket_bytes = [4 * (sum(b.numel() for b in bk["dense"]) + (bk["small"].numel() if bk["small"] is not None else 0)) for bk in self.buckets]

    @staticmethod
    def segmented_backward(forward_loss, P, cut_modules, hops, part, last=None):
        """The three parts of the step (capture-independent: `part(i)` is the context the i-th part runs in -- a graph capture in
        _capture, a null context in the CPU test).  The backward is CUT at the plan's activations: in the forward each of them is
        replaced by a detached leaf (the module's forward hook returns it), so the sub-graphs on either side are disjoint and
        `autograd.grad` over one of them neither needs nor runs the other (without the cut, d loss / d f8 is a TOTAL derivative:
        autograd would run layer3 and layer4 to deliver it).  Every hop has its own parameter group (its own fused-cast node).
        -> ([g0, g1, g2] gradient lists, [groups of P behind each list], tensors to keep alive)."""
        orig, leaf, arrived = {}, {}, {}        # arrived: key -> gradient that has reached that leaf so far (None: nothing yet)
        grad = torch.autograd.grad

        def cut(key):
            def hook(mod, inp, out):
                orig[key] = out
                leaf[key] = out.detach().requires_grad_(True)
                return leaf[key]
            return hook
        hooks = [m.register_forward_hook(cut(key)) for key, m in cut_modules]
        keys = [k for k, _ in cut_modules]

        def walk(hs):
            got = {}
            for src, g_id, dst in hs:
                wrt = P[g_id] + ([leaf[dst]] if dst is not None else [])
                g = grad(orig[src], wrt, grad_outputs=arrived[src], allow_unused=True)
                got[g_id] = list(g[:len(P[g_id])])
                if dst is not None:
                    arrived[dst] = g[-1] if arrived.get(dst) is None else arrived[dst] + g[-1]
            return got
        try:
            with part(0):
                loss = forward_loss()
                g = grad(loss, P[0] + [leaf[k] for k in keys], allow_unused=True)
                g0 = list(g[:len(P[0])])
                for k, gk in zip(keys, g[len(P[0]):]):
                    arrived[k] = gk
                first = dict(arrived)
            lists, groups = [g0], [[0]]
            for i, hs in enumerate(hops):
                with part(i + 1):
                    got = walk(hs)
                    order = sorted(got)
                    lists.append([g_ for k in order for g_ in got[k]])
                    groups.append(order)
                    if i == len(hops) - 1 and last is not None:
                        last()          # last node of the step (the captured step: the next replay draws fresh dropout masks)
        finally:
            for h in hooks:
                h.remove()
        return lists, groups, {"orig": orig, "leaf": leaf, "gb": first, "arrived": arrived}

    @staticmethod
    def build_buckets(grad_lists, device):
        """One all-reduce bucket per segment: the flat fp32 buffers the segment's fused cast wrote its gradients into ("dense": reduced
        in place, no copies) + one flattened remainder for the gradients that are not views of such a buffer ("rest" -> "small")."""
        buckets = []
        for gs in grad_lists:
            bases, members, rest = {}, {}, []
            for g_ in gs:
                b = g_._base
                if b is not None and b.dim() == 1 and b.dtype == torch.float32 and b.is_contiguous():
                    bases[b.data_ptr()] = b
                    members[b.data_ptr()] = members.get(b.data_ptr(), 0) + g_.numel()
                else:
                    rest.append(g_)
            dense = []
            for ptr, b in bases.items():
                if 2 * members[ptr] >= b.numel():
                    dense.append(b)
                else:           # mostly gaps (does not happen with one cast group per segment): its members go one by one
                    rest += [g_ for g_ in gs if g_._base is b]
            small = torch.empty(sum(g_.numel() for g_ in rest), dtype=torch.float32, device=device) if rest else None
            views = [v.view_as(g_) for v, g_ in zip(small.split([g_.numel() for g_ in rest]), rest)] if rest else []
            buckets.append({"dense": dense, "rest": rest, "small": small, "views": views})
        return buckets

    def _reduce_async(self, bk):
        works = []
        avg = dist.get_backend() == "nccl"              # RCCL averages inside the collective; gloo (rehearsal) sums, divided below
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        if bk["small"] is not None:
            torch._foreach_copy_(bk["views"], bk["rest"])
            works.append(dist.all_reduce(bk["small"], op=op, async_op=True))
        for b in bk["dense"]:
            works.append(dist.all_reduce(b, op=op, async_op=True))
        return works

    def replay_and_reduce(self):
        works = []
        for g, bk in zip(self.graphs, self.buckets):
            g.replay()
            if self.world > 1:
                works += self._reduce_async(bk)         # runs on the process group's stream, behind this replay, beside the next one
        self.rng.replayed()
        if self.world > 1:
            for w in works:
                w.wait()
            if dist.get_backend() != "nccl":
                torch._foreach_div_([b for bk in self.buckets for b in bk["dense"]] + [bk["small"] for bk in self.buckets if bk["small"] is not None],
                                    float(self.world))
            for bk in self.buckets:
                if bk["small"] is not None:
                    torch._foreach_copy_(bk["rest"], bk["views"])


def time_msda_kernels(n_frames, device, iters=20, noise=0.0, outliers=0.0, select=True):
    """Live HIP-event timing of the MSDeformAttn kernels at the encoder / decoder shapes of this run, on the launch
    stream (used when the step itself is a graph replay, where per-kernel events cannot be interleaved).  noise / outliers:
    gaussian noise (pixels) on the model's initial ring offsets and a share of samples anywhere in the map -- what trained
    offsets look like to the kernels (only the encoder shape is timed then)."""
    from ocpg_amd.models.ops.functions import ms_deform_attn_func as f
    shapes_l = [((HEIGHT + 31) // 32 * 32 // 8 >> i, (WIDTH + 31) // 32 * 32 // 8 >> i) for i in range(4)]
    shapes = torch.tensor(shapes_l, dtype=torch.long)
    ls = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    sh, lsd = shapes.to(device), ls.to(device)
    sh._ocpg_host = shapes
    g = torch.Generator().manual_seed(0)
    out = {}
    for tag, Lq in ((("enc", S),) if noise or outliers else (("enc", S), ("dec", 5))):
        value = torch.randn(n_frames, S, 8, 32, generator=g).to(device)
        if Lq == S:      # the model's own pattern at init: reference point = the pixel itself, ring offsets of 1..4 px
            import math
            refs = []
            for (h, w) in shapes_l:
                ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
                refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
            ref = torch.cat(refs, 0)[None, :, None, None, None, :]
            th = torch.arange(8) * (2 * math.pi / 8)
            ring = torch.stack([th.cos(), th.sin()], -1)
            ring = (ring / ring.abs().max(-1, keepdim=True)[0]).view(1, 1, 8, 1, 1, 2) * torch.arange(1, 5).view(1, 1, 1, 1, 4, 1)
            norm = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
            loc = (ref + ring / norm).expand(n_frames, S, 8, 4, 4, 2)
            if noise:
                loc = loc + noise * torch.randn(n_frames, S, 8, 4, 4, 2, generator=g) / norm
            if outliers:
                far = torch.rand(n_frames, S, 8, 4, 4, 1, generator=g) < outliers
                loc = torch.where(far, torch.rand(n_frames, S, 8, 4, 4, 2, generator=g) * 1.2 - 0.1, loc)
            loc = loc.contiguous().to(device)
        else:
            loc = torch.rand(n_frames, Lq, 8, 4, 4, 2, generator=g).to(device)
        attn = torch.softmax(torch.randn(n_frames, Lq, 8, 16, generator=g), -1).view(n_frames, Lq, 8, 4, 4).to(device)
        go = torch.randn(n_frames, Lq, 256, generator=g).to(device)
        # one call site's path-selection state (what a module's `_sel_state` buffer is): after two calls on these offsets it has settled
        sel = torch.zeros(8, dtype=torch.int32, device=device) if (Lq == S and select) else None
        for _ in range(3):
            f.ms_deform_attn_forward(value, sh, lsd, loc, attn)
            f.ms_deform_attn_backward(value, sh, lsd, loc, attn, go, sel_state=sel)
        f.enable_kernel_timing(True)
        for _ in range(iters):
            f.ms_deform_attn_forward(value, sh, lsd, loc, attn)
            f.ms_deform_attn_backward(value, sh, lsd, loc, attn, go, sel_state=sel)
        out.update(f.collect_kernel_timing())
        if sel is not None:
            st = sel.tolist()
            out["selection"] = {"path": "tiled" if st[3] else "column", "far_share": st[6] / max(st[7], 1)}
    return out


VALU_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 vector peak (= the fp32-input MFMA peak)
MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)


def _esz(code):
    return 4 if code == 0 else 2


# symbol -> (bound, work of one call from its arguments as passed): algorithmic bytes (each operand once) or FLOPs
LIB_WORK = {
    "ocpg_bn_act_fwd": ("hbm", lambda a: a[5] * a[6] * a[7] * _esz(a[9]) * (2 + bool(a[3]))),
    "ocpg_bn_act_bwd": ("hbm", lambda a: a[5] * a[6] * a[7] * _esz(a[9]) * (2 + bool(a[3]) + bool(a[4] and a[4] != a[0]))),
    "ocpg_dropout_add_ln_fwd": ("hbm", lambda a: a[4] * a[5] * (_esz(a[10]) + 4 + 4)),
    "ocpg_dropout_add_ln_bwd": ("hbm", lambda a: a[6] * a[7] * (4 + _esz(a[11]) + 4 + (_esz(a[11]) if a[12] else 0) + (4 if a[13] else 0))),
    "ocpg_bias_relu_dropout_fwd": ("hbm", lambda a: a[2] * a[3] * _esz(a[7]) * 2),
    "ocpg_bias_relu_dropout_bwd": ("hbm", lambda a: a[2] * a[3] * _esz(a[5]) * 3),
    # MSO (csrc/mso.hip): every map once -- input, output, the optional mask / residual / shared addend; weight gradient: x and g
    "ocpg_mso_conv3x3": ("hbm", lambda a: a[13] * a[14] * a[15] * (a[16] * _esz(a[1]) + a[17] * (_esz(a[12]) + (_esz(a[9]) if a[8] else 0)
                                                                  + (4 if a[10] else 0))) + (a[7] * a[14] * a[15] * a[17] * 4 if a[6] else 0)),
    "ocpg_mso_wgrad": ("hbm", lambda a: a[7] * a[8] * a[9] * (a[10] * _esz(a[1]) + a[11] * 4)),
    "ocpg_colsum_partials": ("hbm", lambda a: a[1] * a[2] * _esz(a[3])),
    "ocpg_im2col3x3_nhwc": ("hbm", lambda a: _esz(a[8]) * a[4] * a[1] * (a[2] * a[3] + 9 * ((a[2] - 1) // a[5] + 1) * ((a[3] - 1) // a[5] + 1))),
    # optimizer: 2048-element chunks (the last chunk of a tensor is partial: an upper bound); p, g, m, v read, p, m, v written / g read
    "ocpg_adamw_step": ("hbm", lambda a: a[9] * 2048 * 28),
    "ocpg_grad_norm_clip": ("hbm", lambda a: a[4] * 2048 * 4),
    "ocpg_dynmask_fwd_f32": ("valu", lambda a: a[3] * a[4] * a[6] * a[7] * (2 * 16 * (a[5] + 2) + 2 * 16 * 16)),
    "ocpg_dynmask_bwd_f32": ("valu", lambda a: 2 * a[3] * a[4] * a[6] * a[7] * (2 * 16 * (a[5] + 2) + 2 * 16 * 16)),
    # matrix-core kernels: FLOPs = 2 * M * N * K (* batch); bf16 / fp16 storage -> the 2.5 PF peak, fp32 GEMMs -> the 157 TF fp32 peak
    "ocpg_gemm": ("mfma", lambda a: 2.0 * a[8] * a[9] * a[10] * max(a[14], 1), lambda a: a[4] != 0),
    "ocpg_gemm_bn_act": ("mfma", lambda a: 2.0 * a[8] * a[9] * a[10], lambda a: a[7] != 0),
    "ocpg_conv3x3_mfma_fwd": ("mfma", lambda a: 2.0 * a[5] * ((a[6] - 1) // a[10] + 1) * ((a[7] - 1) // a[10] + 1) * 9 * a[8] * a[9], lambda a: True),
    "ocpg_conv3x3_mfma_fwd_cols": ("mfma", lambda a: 2.0 * a[5] * ((a[6] - 1) // a[10] + 1) * ((a[7] - 1) // a[10] + 1) * 9 * a[8] * a[9], lambda a: True),
    "ocpg_conv3x3_mfma_dgrad": ("mfma", lambda a: 2.0 * a[2] * ((a[3] - 1) // a[7] + 1) * ((a[4] - 1) // a[7] + 1) * 9 * a[5] * a[6], lambda a: True),
    "ocpg_small_linear_fwd": ("mfma", lambda a: 2.0 * a[4] * a[5] * a[6], lambda a: True),
    "ocpg_small_linear_bwd": ("mfma", lambda a: 4.0 * a[6] * a[7] * a[8], lambda a: True),
    "ocpg_win_attn_fwd": ("mfma", lambda a: 4.0 * a[4] * a[7] * a[6] * a[6] * a[8], lambda a: a[11] != 0),
    "ocpg_win_attn_bwd_mfma": ("mfma", lambda a: 10.0 * a[5] * a[8] * a[7] * a[7] * a[9], lambda a: True),
}


def lib_work(name, args):
    """Work of one call (algorithmic bytes or FLOPs); for matrix-core symbols a pair (FLOPs, low-precision operands?)."""
    f = LIB_WORK.get(name)
    try:
        if not f:
            return None
        if f[0] == "mfma":
            return float(f[1](args)), bool(f[2](args))
        return float(f[1](args))
    except (TypeError, IndexError):
        return None


def kernel_table(msda_kt, lib_kt, n_frames, steps, lib_steps):
    """Per-kernel roofline rows from the live event timings.  Algorithmic bytes / FLOPs per launch follow DESIGN.md section 4
    (what the kernel must read and write once; fp32 unless the call says bf16)."""
    hp, wp = (HEIGHT + 31) // 32 * 32, (WIDTH + 31) // 32 * 32
    S, M, D, LP = sum((hp // 8 >> i) * (wp // 8 >> i) for i in range(4)), 8, 32, 16
    v, la, o = 4 * n_frames * S * M * D, 4 * n_frames * 3 * S * M * LP, 4 * n_frames * S * M * D     # value, loc+attn, out bytes
    msda_bytes = {"fwd_enc": v + la + o,                    # read value, loc, attn; write out
                  "bwd_enc_value": la + o + v,              # read loc, attn, grad_out; write grad_value
                  "bwd_enc_locattn": v + la + o + la,       # read value, loc, attn, grad_out; write grad_loc, grad_attn
                  "bwd_enc": v + la + o + v + la}           # the whole backward when one call serves it
    rows = []
    for k, d in msda_kt.items():
        if d["n"]:
            us = d["ms"] / d["n"] * 1e3
            row = {"kernel": "msda_" + k, "us": us, "launches_per_step": d["n"] / steps}
            if k in msda_bytes:
                row.update(bound="hbm", algorithmic_bytes=msda_bytes[k], achieved_GBs=msda_bytes[k] / us / 1e3,
                           frac=msda_bytes[k] / us / 1e3 / HBM_PEAK_GBS)
            rows.append(row)

    for k, d in lib_kt.items():
        if not d["n"]:
            continue
        us = d["ms"] / d["n"] * 1e3
        row = {"kernel": k, "us": us, "launches_per_step": d["n"] / lib_steps}
        if d["modelled"] == d["n"] and d["work"] > 0:
            q = d["work"] / d["n"]
            if LIB_WORK[k][0] == "hbm":
                row.update(bound="hbm", algorithmic_bytes=q, achieved_GBs=q / us / 1e3, frac=q / us / 1e3 / HBM_PEAK_GBS)
            elif LIB_WORK[k][0] == "mfma":
                # calls of one symbol mix bf16 and fp32 operands (ocpg_gemm): the peak is weighted by each call's FLOPs
                peak = (d.get("work_lowp", 0.0) * MFMA_BF16_PEAK_TFLOPS + (d["work"] - d.get("work_lowp", 0.0)) * VALU_PEAK_TFLOPS) / d["work"]
                row.update(bound="mfma", flops=q, achieved_TFLOPs=q / us / 1e6, peak_TFLOPs=peak, frac=q / us / 1e6 / peak,
                           lowp_flop_share=d.get("work_lowp", 0.0) / d["work"])
            else:
                row.update(bound="valu_fp32", flops=q, achieved_TFLOPs=q / us / 1e6, frac=q / us / 1e6 / VALU_PEAK_TFLOPS)
        rows.append(row)
    rows.sort(key=lambda r: -r["us"] * r["launches_per_step"])
    return rows


GV_KERNELS = {"column": ("k_scatter_col4",), "tiled": ("k_gv_tile", "k_gv_coarse")}       # grad_value kernel family -> kernel names (csrc/msda_col.hip, msda_tile.hip)


def pmc_traffic(kernels, offsets, n_frames):
    """HBM bytes per launch (sum over `kernels`) from the committed rocprofv3 --pmc passes, profiles/r04_msda_pmc.json, keyed by the
    LAUNCHED kernels' names and the offset pattern -- collected and corrected as MI355X_MICROARCH.md prescribes (tools/pmc_gv.sh: one
    counter per pass; reads = 2 x FETCH_SIZE on gfx950).  No file: None.  A file that lacks one of the kernels, or was taken at another
    shape, is an ERROR: a stale file must never label a different kernel (round 3 read round 2's key)."""
    path = os.path.join(ROOT, "profiles", "r04_msda_pmc.json")
    if not os.path.exists(path):
        return None
    pmc = json.load(open(path))
    if pmc.get("n_frames") != n_frames or (HEIGHT, WIDTH) != (384, 640):
        return None
    rows = pmc["offsets"][offsets]
    return float(sum(rows[k]["hbm_bytes_per_launch"] for k in kernels))


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never will) and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's (and torch's) sharing of
    # device memory between the ranks' processes fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it already; the
    # child environment states it so that a launch from a stripped environment behaves the same.  (An explicit setting is kept.)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def choose_step(rungs, world, agree_device, reset, trial_steps=3, needs_finite=True, log=None):
    """The launch-mode LADDER: `rungs` = [(label, build)] in order of preference (three overlapped graphs -> one graph -> ...); build()
    constructs and checks a step object (may raise).  After every attempt ALL ranks vote (MIN all-reduce): a rung counts only if it
    worked on every rank -- the modes differ in their collectives (bucketed async all-reduces / one flat all-reduce / DDP's hooks), so
    the ranks must never disagree, and a rank that succeeded alone drops its step and follows.  Then `trial_steps` real steps (they
    contain the gradient collectives: every rank runs all of them) and a second vote on their outcome.  reset(): back to the start
    state after a failed rung.  Returns (label, step) of the first rung that survived, or (None, None): the caller runs eagerly.
    A failure never re-launches anything in this process: it degrades one rung, and a hang inside a collective ends at the process
    group's timeout with a non-zero exit."""
    log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))

    def vote(ok):
        if world <= 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=agree_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))
    for label, build in rungs:
        step, ok = None, True
        try:
            step = build()
        except Exception as e:      # capture is an optimisation, never a requirement
            log(f"[bench] launch mode '{label}' failed here ({type(e).__name__}: {str(e)[:300]})")
            ok = False
        if not vote(ok):
            if ok:
                log(f"[bench] launch mode '{label}' failed on another rank: following")
            step = None
            reset()
            continue
        fine = True
        try:
            for _ in range(trial_steps):
                loss = step()
                if needs_finite and not bool(torch.isfinite(loss)):
                    fine = False
        except Exception as e:
            log(f"[bench] launch mode '{label}': trial step failed ({type(e).__name__}: {str(e)[:200]})")
            fine = False
        if vote(fine):
            return label, step
        log(f"[bench] launch mode '{label}' did not survive its trial steps (on some rank): next rung")
        step = None
        reset()
    return None, None


def cpu_baseline(state_shapes):
    """Time the CPU oracle (a restatement of the reference path) on ONE clip: forward + criterion + backward."""
    try:
        from oracle import ocpg_ref
    except ImportError:
        return None
    return ocpg_ref.timed_baseline(model_args("cpu", amp=False), state_shapes, T_FRAMES, HEIGHT, WIDTH)


def main():
    global T_FRAMES, HEIGHT, WIDTH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips-per-gpu", type=int, default=2)
    ap.add_argument("--backbone", default="resnet101")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--frames", type=int, default=T_FRAMES)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--text", default="features", choices=["features", "roberta"],
                    help="features: random [B,9,768] text features (configs #1-#4); roberta: captions through RoBERTa-base (config #5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-b1", action="store_true", help="skip the extra B = 1 leg (1 clip per GPU and step) after the timed run")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel eagerly (DistributedDataParallel for N > 1).  Default: forward + criterion + backward "
                         "replayed as ONE HIP graph (the eager step is host-bound: ~2 600 launches, 45 ms vs 40 ms); the graph is "
                         "checked against an eager step before the timed region and bench.py falls back to eager if that fails.")
    ap.add_argument("--graph", action="store_true", help="(default since round 2; kept for old command lines)")
    a = ap.parse_args()
    T_FRAMES, HEIGHT, WIDTH = a.frames, a.height, a.width

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # OCPG_REHEARSE_ONE_GPU=1: rehearsal of the N>1 code path on a one-GPU box (all ranks on cuda:0, gloo instead of RCCL,
    # which refuses two ranks per device); never a measurement
    rehearse = os.environ.get("OCPG_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        # a rank that dies (or leaves a collective) must not hang the others for NCCL's default 10 minutes: the run fails, non-zero, soon
        tmo = datetime.timedelta(seconds=int(os.environ.get("OCPG_DIST_TIMEOUT", "300")))
        if rehearse:
            dist.init_process_group(backend="gloo", init_method="env://", timeout=tmo)
        else:
            dist.init_process_group(backend="nccl", init_method="env://", device_id=device, timeout=tmo)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"

    from ocpg_amd import _lib
    _lib.lib()          # fail loudly if the HIP extension is missing
    from ocpg_amd.models import build_model
    from ocpg_amd.models.ops.functions import ms_deform_attn_func as msda_fn

    torch.manual_seed(42)           # replicas start from the SAME weights (the graph step only all-reduces gradients) ...
    torch.backends.cudnn.benchmark = True
    if os.environ.get("OCPG_BLAS"):          # A/B: 'hipblas' (rocBLAS) | 'hipblaslt'
        torch.backends.cuda.preferred_blas_library(os.environ["OCPG_BLAS"])
    args = model_args(device, a.backbone, amp=a.dtype != "fp32", roberta=a.text == "roberta")
    model, criterion, _ = build_model(args)
    model.to(device)
    for m in model.modules():       # NHWC weights for the 2-D convs (MIOpen's fast bf16 kernels); 3-D/other modules untouched
        if isinstance(m, torch.nn.Conv2d):
            m.to(memory_format=torch.channels_last)
    criterion.to(device)
    model.train(), criterion.train()
    if world > 1:                   # ... and by construction: rank 0's parameters and buffers everywhere (what DDP's constructor does)
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t, src=0)
    torch.manual_seed(42 + rank)    # per-rank stream for data and dropout only (main.py:44-45)
    optimizer = make_optimizer(model, args)
    amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[a.dtype]
    make_samples, text, targets = synthetic_batch(a.clips_per_gpu, device, seed=42 + rank, roberta=a.text == "roberta")
    step, mode = None, "eager"
    gemm_routing = ("hipBLASLt plan cache (ocpg_gemm: " + ("the heuristic's ranked dozen" if os.environ.get("OCPG_GEMM_TUNE_ALL") == "0"
                                                           else "every Tensile kernel that supports the problem") +
                    " timed and validated at first use) for every GEMM-shaped layer")
    eager_routing = "at::mm outside the ResNet body (the eager step is host-bound: DESIGN.md section 5)"
    fallback_note = None
    if a.eager and "OCPG_PLANNED_GEMM" not in os.environ:
        # the eager step is bound by the host, and with every GEMM routed through the plan cache it is slower (measured: 55.9 ms
        # against 45.3 ms with at::mm outside the ResNet body); graph replays have no host cost
        from ocpg_amd.models.ops.functions import gemm_func
        gemm_func.PLANNED = False
        gemm_routing = eager_routing
    if world > 1 and rank != 0:
        from ocpg_amd.util import gemm_sync
        gemm_sync.follow()              # rank 0 times the GEMM candidates; the others import its choices below (same kernels on every rank)
    shared_picks = None
    if not a.eager:
        snapshot = {k: v.clone() for k, v in model.state_dict().items()}
        nonlocal_state = {"optimizer": optimizer}

        def back_to_start():
            torch.cuda.synchronize()
            from ocpg_amd.models import amp_cache
            amp_cache._PARTIALS.clear()         # an aborted backward may leave deferred partial sums registered
            amp_cache.set_groups(model, None)
            nonlocal_state["optimizer"] = make_optimizer(model, args)
            model.load_state_dict(snapshot)                      # failed replays may have poisoned the weights
            criterion.iter_device, criterion.iter = None, 0
            try:
                from ocpg_amd.models.matcher import raise_if_malformed_boxes
                raise_if_malformed_boxes()
            except AssertionError:
                pass
        torch.manual_seed(1234 + rank)
        model.zero_grad(set_to_none=True)
        eager_loss = float(forward_backward(model, criterion, make_samples(), text, targets, amp_dtype))
        model.zero_grad(set_to_none=True)
        criterion.iter = 0
        if world > 1:
            from ocpg_amd.util import gemm_sync
            shared_picks = gemm_sync.share(device)       # every GEMM shape of the step was seen (and timed on rank 0) in that eager step
        seg = os.environ.get("OCPG_GRAPH_SEGMENTS", "auto")      # auto: three graphs with overlapped all-reduces when N > 1
        use_seg = SegmentedGraphStep.supported(model, amp_dtype) and (seg == "3" or (seg == "auto" and world > 1))

        def build(cls):
            def f():
                st = cls(model, criterion, nonlocal_state["optimizer"], make_samples, text, targets, args, amp_dtype, world)
                st.check(eager_loss)                             # one replay against an eager step on the same masks, no collective
                return st
            return f
        rungs = ([("hipgraph(3 segments: bucket i all-reduced while graph i+1 replays)", build(SegmentedGraphStep))] if use_seg else []) + \
                [("hipgraph(fwd+criterion+bwd)", build(GraphStep))]
        label, step = choose_step(rungs, world, device, back_to_start, needs_finite=amp_dtype != torch.float16)   # fp16: overflowed steps are the scaler's to skip
        optimizer = nonlocal_state["optimizer"]
        if step is not None:
            mode = label
        else:
            fallback_note = "every captured launch mode failed (see stderr): eager launches"
            print("[bench] " + fallback_note, file=sys.stderr, flush=True)
            if "OCPG_PLANNED_GEMM" not in os.environ:          # same host-cost trade as --eager (above): a DIFFERENT GEMM routing, labelled in the line
                from ocpg_amd.models.ops.functions import gemm_func
                gemm_func.PLANNED = False
                gemm_routing = eager_routing
    if step is None:
        ddp_model = model
        if world > 1:
            ddp_model = wrap_ddp(model, local_rank)
        step = EagerStep(model, ddp_model, criterion, optimizer, make_samples, text, targets, args, amp_dtype)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    if not a.no_kernel_timing and mode == "eager":
        msda_fn.enable_kernel_timing(True)          # 24 event pairs per step: no measurable host cost
    t0 = time.perf_counter()
    losses = []
    for _ in range(a.steps):
        loss = step()
        losses.append(loss.clone())
    sync()
    dt = time.perf_counter() - t0
    from ocpg_amd.models.matcher import raise_if_malformed_boxes
    try:
        raise_if_malformed_boxes()
    except AssertionError as e:
        if a.dtype != "fp16":
            raise
        print(f"[bench] fp16 run: {e} (overflowed steps are skipped by the GradScaler)", file=sys.stderr)
    if a.dtype != "fp16" and not all(bool(torch.isfinite(l)) for l in losses):     # fp16: the GradScaler skips overflowed steps
        raise AssertionError("non-finite loss during the timed steps: " + " ".join(f"{float(l):.2f}" for l in losses))
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    per_rank = [dt]
    if world > 1:
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank = [float(x.item()) for x in every]          # a straggler (a rank on slower kernels / a slower host) shows here
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    kt = msda_fn.collect_kernel_timing() if (not a.no_kernel_timing and mode == "eager") else {}
    lib_kt, lib_steps, kt_steps = {}, 3, a.steps
    if not a.no_kernel_timing and rank == 0 and world == 1:
        # per-kernel HIP-event timing of every library call over a few EXTRA eager steps outside the timed region, same
        # process, same tensors (an event pair around each of ~500 calls per step costs host time: +20 ms per eager step; a
        # graph replay cannot carry events at all, so in graph mode the MSDeformAttn kernels are timed here as well)
        estep = step if mode == "eager" else EagerStep(model, model, criterion, optimizer, make_samples, text, targets, args, amp_dtype)
        if mode != "eager":
            msda_fn.enable_kernel_timing(True)
        _lib.enable_kernel_timing(True)
        for _ in range(lib_steps):
            estep()
        lib_kt = _lib.collect_kernel_timing(lib_work)
        if mode != "eager":
            kt, kt_steps = msda_fn.collect_kernel_timing(), lib_steps
            msda_fn.enable_kernel_timing(False)
    elif not a.no_kernel_timing and mode != "eager" and rank == 0:
        kt = time_msda_kernels(a.clips_per_gpu * T_FRAMES, device)       # N > 1: the other ranks cannot join extra eager steps

    clips = a.steps * a.clips_per_gpu * world
    line = {
        "metric": "clips/sec fwd+bwd (5x384x640, R101)" if (a.backbone, a.frames, a.height, a.width) == ("resnet101", 5, 384, 640)
        else f"clips/sec fwd+bwd ({a.frames}x{a.height}x{a.width}, {a.backbone})", "value": clips / dt, "unit": "clips/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{a.backbone} + 4-scale deformable transformer (4 enc + 4 dec), {T_FRAMES}x{HEIGHT}x{WIDTH} clips, "
                               f"5 queries, {a.clips_per_gpu} clips/GPU/step, step = fwd + criterion + bwd + clip + AdamW",
                   "global_batch": a.clips_per_gpu * world, "parallelism": f"dp{world}", "weights": "random init",
                   "text": "random features [B,9,768] (RoBERTa bypassed, BASELINE configs #1-#4)" if a.text == "features"
                           else "10-word captions through a random-init RoBERTa-base (frozen)", "launch": mode,
                   "gemm_routing": gemm_routing,
                   "matcher_bit_exact": "fp32 mode only (under 16-bit autocast a near-tied assignment can move: tests/test_model_gpu.py::test_full_size_step_vs_oracle)"},
        "final_loss": float(loss.detach()),
    }
    if fallback_note:
        line["config"]["launch_fallback"] = fallback_note
    if getattr(step, "phases", None):
        line["step_phases_ms"] = {k: v / (a.steps + a.warmup + 3) * 1e3 for k, v in step.phases.items()}
    if mode != "eager":
        line["hipgraph"] = {"memset_nodes_replaced_by_kernel_nodes": step.memset_nodes_replaced,
                            "checked_against_eager": getattr(step, "check_result", True), "graphs": len(step.graphs)}
        if hasattr(step, "bucket_bytes"):
            line["hipgraph"]["allreduce_buckets_bytes"] = step.bucket_bytes       # bucket i is reduced while graph i + 1 replays
    line["ranks"] = {"world_size": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None,
                     "ms_per_step_min": min(per_rank) / a.steps * 1e3, "ms_per_step_max": max(per_rank) / a.steps * 1e3,
                     "gemm_choices_shared_from_rank0": shared_picks}
    import ctypes
    changed = ctypes.c_longlong(0)
    line["gemm_plans"] = {"cached": int(_lib.lib().ocpg_gemm_plans()), "timed_at_first_use": int(_lib.lib().ocpg_gemm_tuned(ctypes.addressof(changed))),
                          "left_first_heuristic_choice": int(changed.value),
                          "candidates_rejected_for_differing_result": int(_lib.lib().ocpg_gemm_tune_rejected())}
    if rank == 0:
        rows = kernel_table(kt, lib_kt, a.clips_per_gpu * T_FRAMES, kt_steps, lib_steps)
        dom = next((r for r in rows if r["kernel"] in ("msda_bwd_enc_value", "msda_bwd_enc") and "frac" in r), None)
        if dom is not None:
            nfr = a.clips_per_gpu * T_FRAMES
            what = "grad_value of the MSDeformAttn backward (ocpg_msda_bwd_value_sel_f32), encoder shape, N=%d frames" % nfr
            # what the STEP ran: the model's initial ring offsets (random-init weights) -> the column scatter, timed live in this process
            ring = {"bound": "hbm", "kernel": "k_scatter_col4 (+ the idle launches of the tiled family and the commit) -- " + what,
                    "offsets": "the step's own: initial ring (random-init weights)",
                    "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                    "traffic": pmc_traffic(GV_KERNELS["column"], "ring", nfr),
                    "launch_us": dom["us"], "algorithmic_bytes": dom["algorithmic_bytes"],
                    "launches_timed": int(round(dom["launches_per_step"] * kt_steps)),
                    "timed_in": "the timed steps" if mode == "eager" else f"{kt_steps} extra eager steps after the timed (graph) steps"}
            line["roofline"] = ring
            if world == 1 and not a.no_kernel_timing and (HEIGHT, WIDTH) == (384, 640):
                # the ring row once more with the calls issued back to back (the extra eager steps are host-bound: the four launches of
                # the selecting entry point -- active kernel, two idle ones, commit -- show their launch gaps there, a graph replay does not)
                rk = time_msda_kernels(nfr, device).get("bwd_enc_value")
                # ... and the KERNEL ALONE: the plain entry point (ocpg_msda_bwd_value_f32) launches k_scatter_col4 and nothing else -- the
                # duration rocprofv3's per-kernel average of the same command has to agree with (profiles/r04_bench_rocprofv3_kernel_stats.csv)
                r1 = time_msda_kernels(nfr, device, select=False).get("bwd_enc_value")
                if rk and rk["n"] and r1 and r1["n"]:
                    us0, us1 = rk["ms"] / rk["n"] * 1e3, r1["ms"] / r1["n"] * 1e3
                    ring.update(kernel="k_scatter_col4 -- " + what, launch_us_in_eager_steps=ring["launch_us"], entry_point_us=us0, launch_us=us1,
                                achieved=dom["algorithmic_bytes"] / us1 / 1e3, frac=dom["algorithmic_bytes"] / us1 / 1e3 / HBM_PEAK_GBS,
                                launches_timed=r1["n"],
                                timed_in="this process, after the timed steps, calls back to back on the model's initial ring offsets (HIP events on "
                                         "the launch stream): launch_us = the kernel alone (ocpg_msda_bwd_value_f32: one launch), entry_point_us = the "
                                         "selecting entry point the step calls (this kernel + two idle launches of the tiled family + the one-thread "
                                         "commit), launch_us_in_eager_steps = that entry point inside the extra eager steps")
                # `roofline` stays the kernel the timed steps launch (the contract's definition); next to it the same entry point
                # on perturbed ("trained-like") offsets -- +3 px gaussian noise, 5 % of the samples anywhere in the map -- through the
                # per-call selection, which moves such a call site to the output-tiled kernels
                tk = time_msda_kernels(nfr, device, noise=3.0, outliers=0.05)
                pk, selinfo = tk.get("bwd_enc_value"), tk.get("selection", {"path": "column", "far_share": None})
                if pk and pk["n"]:
                    us = pk["ms"] / pk["n"] * 1e3
                    fam = selinfo["path"]
                    line["roofline_trained_like_offsets"] = {"bound": "hbm", "kernel": " + ".join(GV_KERNELS[fam]) + f" (selected: {fam} family) -- " + what,
                                        "offsets": "ring + N(0, 3 px) + 5 % uniform (trained-like)", "far_share_seen": selinfo["far_share"],
                                        "achieved": dom["algorithmic_bytes"] / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": dom["algorithmic_bytes"] / us / 1e3 / HBM_PEAK_GBS,
                                        "traffic": pmc_traffic(GV_KERNELS[fam], "trained", nfr), "launch_us": us,
                                        "algorithmic_bytes": dom["algorithmic_bytes"], "launches_timed": pk["n"],
                                        "timed_in": "this process, after the timed steps: the entry point on synthetic trained-like offsets (HIP events on the launch stream)"}
        if rows:
            line["kernels"] = rows[:16]
        from ocpg_amd.models import fallbacks
        line["library_fallbacks"] = fallbacks.snapshot()         # attention calls that left the HIP kernels for torch SDPA: {} at config #2
        if world == 1 and mode != "eager" and not a.no_b1 and a.clips_per_gpu != 1:
            # BASELINE config #2 as SURVEY section 8 words it (B = 1 clip per GPU and step) from the same process; the headline `value`
            # is the throughput configuration (2 clips per GPU and step = config #3's per-GPU batch)
            try:
                mk1, text1, targets1 = synthetic_batch(1, device, seed=42 + rank, roberta=a.text == "roberta")
                step1 = GraphStep(model, criterion, optimizer, mk1, text1, targets1, args, amp_dtype, world)
                for _ in range(3):
                    step1()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    l1 = step1()
                torch.cuda.synchronize()
                d1 = time.perf_counter() - t1
                line["b1"] = {"clips_per_gpu": 1, "value": a.steps / d1, "unit": "clips/s", "ms_per_step": d1 / a.steps * 1e3, "steps": a.steps,
                              "final_loss": float(l1.detach())}
            except Exception as e:      # never lose the headline line to the extra leg
                line["b1"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
        if world == 1 and not a.no_cpu_baseline and a.backbone.startswith("resnet"):
            line["cpu_baseline"] = cpu_baseline({k: tuple(v.shape) for k, v in model.state_dict().items() if v.dtype.is_floating_point})
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
