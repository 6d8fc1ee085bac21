"""N>1 path on CPU: 2 ranks over gloo, DistributedDataParallel around the product model (MSDA op -> test double),
each rank a different clip.  Checks (a) DDP-averaged gradients equal the average of the two single-process gradients
(which also exercises the criterion's num_boxes all-reduce + /world_size), (b) every trainable parameter received a
gradient, so find_unused_parameters=False (as bench.py uses) is legitimate; test_bench_eager_step_two_ranks drives
bench.py's OWN step object (wrap_ddp's bucket-view gradients, make_optimizer's four LR groups, clip, AdamW) and checks
the post-step parameters against a single-process replay; test_num_boxes_allreduce covers unequal valid-frame counts
across ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build(meta):
    import model_checks
    import ocpg_amd.models.ops.modules.ms_deform_attn as mod
    from oracle.msda import MSDAOracleFunction
    mod.MSDeformAttnFunction = MSDAOracleFunction
    return model_checks.build_product(meta, torch.device("cpu"))


def _clip_grads(rank_clip, meta, model, crit, ddp=None):
    import cases
    import model_checks
    from ocpg_amd.util.misc import NestedTensor
    T, H, W = meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(2, T, H, W, meta["nopad_sizes"])
    x, mask, targets = x[rank_clip:rank_clip + 1], mask[rank_clip:rank_clip + 1], targets[rank_clip:rank_clip + 1]
    f, s, m = cases.tiny_text(2)
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    text = PrecomputedText(f[rank_clip:rank_clip + 1], s[rank_clip:rank_clip + 1], m[rank_clip:rank_clip + 1])
    net = ddp or model
    model.train(), crit.train()
    crit.iter = 0            # the level-set warm-up weight depends on the criterion's call counter
    model.zero_grad(set_to_none=True)
    out = net(NestedTensor(x, mask), text, targets)
    losses, *_ = crit(out, targets)
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    return {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}, total.item()


def _worker(rank, world, port, q):
    _setup_paths()
    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import Golden
    meta = Golden("e2e_tiny").meta
    _, model, crit = _build(meta)
    ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=False)
    grads, total = _clip_grads(rank, meta, model, crit, ddp)
    q.put((rank, {k: (None if g is None else g.numpy()) for k, g in grads.items()}, total))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_gloo():
    _setup_paths()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, g, t = q.get(timeout=500)
        got[r] = (g, t)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # DDP leaves identical (averaged) grads on both ranks
    for k in got[0][0]:
        a, b = got[0][0][k], got[1][0][k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert abs(a - b).max() <= 1e-6 * (abs(a).max() + 1e-12) + 1e-9, k
    # single-process reference: the same two clips one after the other; DDP must have left their AVERAGE
    from conftest import Golden
    meta = Golden("e2e_tiny").meta
    _, model, crit = _build(meta)
    singles = [_clip_grads(r, meta, model, crit)[0] for r in range(2)]
    n_checked = 0
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        a = got[0][0][k]
        assert a is not None, f"{k} got no gradient under DDP (find_unused_parameters=False would break)"
        want = 0.5 * (singles[0][k] + singles[1][k]).numpy()
        assert abs(a - want).max() <= 2e-4 * (abs(want).max() + 1e-12) + 1e-7, k
        n_checked += 1
    assert n_checked > 100


def _bench_inputs(rank_clip, meta):
    import cases
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    from ocpg_amd.util.misc import NestedTensor
    T, H, W = meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(2, T, H, W, meta["nopad_sizes"])
    sl = slice(rank_clip, rank_clip + 1)
    f, s, m = cases.tiny_text(2)
    return (lambda: NestedTensor(x[sl].clone(), mask[sl].clone())), PrecomputedText(f[sl], s[sl], m[sl]), targets[sl]


def _bench_worker(rank, world, port, q):
    _setup_paths()
    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from conftest import Golden
    meta = Golden("e2e_tiny").meta
    args, model, crit = _build(meta)
    model.train(), crit.train()
    crit.iter = 0
    ddp = bench.wrap_ddp(model)                                  # the wrapper bench.py builds for --gpus N
    opt = bench.make_optimizer(model, args, fused=False)
    make_samples, text, targets = _bench_inputs(rank, meta)
    step = bench.EagerStep(model, ddp, crit, opt, make_samples, text, targets, args, None)
    loss = step()
    # gradient_as_bucket_view: .grad tensors are views of the reducer's buckets and hold the AVERAGED, clipped gradients
    q.put((rank, float(loss), float(step.grad_norm), {k: p.detach().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bench_eager_step_two_ranks():
    """bench.py's EagerStep under a 2-rank gloo group == averaging the two clips' gradients in one process, clipping at
    clip_max_norm and stepping the same four-group AdamW: post-step parameters identical on both ranks and equal to the replay."""
    _setup_paths()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, loss, gn, params = q.get(timeout=500)
        got[r] = (loss, gn, params)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert abs(got[0][1] - got[1][1]) <= 1e-5 * abs(got[0][1])            # same (global) gradient norm on both ranks
    for k, a in got[0][2].items():
        assert abs(a - got[1][2][k]).max() <= 1e-7 * (abs(a).max() + 1e-12) + 1e-10, k
    # single-process replay
    import bench
    from conftest import Golden
    meta = Golden("e2e_tiny").meta
    args, model, crit = _build(meta)
    singles = [_clip_grads(r, meta, model, crit)[0] for r in range(2)]
    opt = bench.make_optimizer(model, args, fused=False)
    for k, p in model.named_parameters():
        if p.requires_grad:
            p.grad = 0.5 * (singles[0][k] + singles[1][k])
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_max_norm, error_if_nonfinite=False)
    assert abs(float(norm) - got[0][1]) <= 2e-4 * float(norm), (float(norm), got[0][1])
    opt.step()
    bad, n_el, n_off = [], 0, 0
    for k, p in model.named_parameters():
        if p.requires_grad:
            want = p.detach().numpy()
            d = abs(got[0][2][k] - want)
            # AdamW's first step moves every element by ~lr * g / (|g| + eps): elements whose averaged gradient is at rounding
            # level (|g| ~ eps = 1e-8) may land anywhere within +-lr; everything else must agree to a few 1e-7
            off = (d > 2e-6)
            n_el += d.size
            n_off += int(off.sum())
            if d.max() > 2.1e-4 or off.mean() > 0.02:
                bad.append((k, float(d.max()), float(off.mean())))
    assert not bad, bad[:5]
    assert n_off <= 2e-3 * n_el, (n_off, n_el)


def _nb_worker(rank, world, port, q):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import module_checks as mc
    from conftest import Golden
    g = Golden("matcher_crit")
    # rank 0 has 3 valid frames in its batch, rank 1 has 1 -> num_boxes = (3 + 1) / 2 = 2 on both
    import cases, synth
    from ocpg_amd.models import build_model
    m = g.meta
    _, crit, _ = build_model(cases.default_args(**m["cfg"]))
    b, t, H, W = m["b"], m["t"], m["H"], m["W"]
    targets = synth.synthetic_targets(b, t, H, W)
    targets[1]["valid"] = torch.tensor([1, 0]) if rank == 0 else torch.tensor([0, 0])
    targets[0]["valid"] = torch.tensor([1, 1]) if rank == 0 else torch.tensor([1, 0])
    out = {"pred_logits": synth.rand("mc_logits", (b, t, m["q"], 1)), "pred_boxes": synth.rand("mc_boxes", (b, t, m["q"], 4), uniform=True) * 0.5 + 0.2}
    idx = [(torch.tensor([0]), torch.tensor([0])) for _ in range(b)]
    nb = torch.stack([tt["valid"] for tt in targets]).sum().float().reshape(1)
    dist.all_reduce(nb)
    expect = (nb / world).clamp(min=1)[0]
    crit.losses = ["boxes"]
    out.update(main_matcher_index=idx, aux_matcher_index=[], pred_masks_low=out["pred_logits"])
    losses, *_ = crit(out, targets)
    sel = out["pred_boxes"][torch.arange(b), :, 0].reshape(-1, 4)
    tb = torch.cat([tt["boxes"] for tt in targets])
    q.put((rank, float(losses["loss_bbox"]), float((sel - tb).abs().sum() / expect), float(expect)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_num_boxes_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nb_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, want, nb in res:
        assert nb == 2.0
        assert abs(got - want) <= 1e-6 * abs(want)


class _FakeGraph:
    """CPU stand-in for one captured segment: a replay refills the segment's STATIC gradient tensors in place."""

    def __init__(self, fill):
        self.fill = fill

    def replay(self):
        self.fill()


def _seg_worker(rank, world, port, q):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    g = torch.Generator().manual_seed(7)
    shapes = [[(4, 3), (5,), (2, 2, 2)], [(6,), (3, 3)], [(7, 2)]]             # three segments of "parameters"
    loose = [[(3,)], [], [(2, 5), (1,)]]                                        # gradients that are NOT views of a flat cast buffer
    values = [[torch.randn(s, generator=g) for s in seg + lo] for seg, lo in zip(shapes, loose)]      # identical on both ranks
    grad_lists, fills = [], []
    for seg, lo, vals in zip(shapes, loose, values):
        n = sum(torch.Size(s).numel() + 3 for s in seg)                         # gaps between the views, like the 16-byte alignment of the plan
        flat = torch.full((n,), float("nan"))                                   # gaps hold garbage (torch.empty in the product)
        views, off = [], 0
        for s in seg:
            k = torch.Size(s).numel()
            views.append(flat[off:off + k].view(s))
            off += k + 3
        singles = [torch.empty(s) for s in lo]
        gl = views + singles
        grad_lists.append(gl)

        def fill(gl=gl, vals=vals):
            for t, v in zip(gl, vals):
                t.copy_(v * (rank + 1))                                          # rank r's gradient = (r + 1) * v  ->  average = 1.5 v
        fills.append(fill)
    step = object.__new__(bench.SegmentedGraphStep)
    step.world = world
    step.graphs = [_FakeGraph(f) for f in fills]
    step.buckets = bench.SegmentedGraphStep.build_buckets(grad_lists, torch.device("cpu"))

    class _Rng:
        def replayed(self):
            pass
    step.rng = _Rng()
    assert [len(b["dense"]) for b in step.buckets] == [1, 1, 1] and [len(b["rest"]) for b in step.buckets] == [1, 0, 2]
    for _ in range(2):                                                           # two steps: the buffers are reused
        step.replay_and_reduce()
    err = max(float((t - 1.5 * v).abs().max()) for gl, vals in zip(grad_lists, values) for t, v in zip(gl, vals))
    q.put((rank, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_segmented_graph_step_reduce_path_two_ranks():
    """bench.py's SegmentedGraphStep (three captured graphs, bucket i all-reduced while graph i + 1 replays): its OWN bucket builder
    and replay_and_reduce over a 2-rank gloo group, with CPU stand-ins for the replays -- every gradient (views of the flat cast
    buffers and loose ones) ends as the average over the ranks, twice in a row."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err in res:
        assert err <= 1e-6, (rank, err)


def _ladder_worker(rank, world, port, q, scenario):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    import bench
    resets, built, steps = [], [], []

    class Step:
        def __init__(self, name, fail_step_on=None):
            self.name, self.fail_step_on = name, fail_step_on

        def __call__(self):
            t = torch.ones(1)
            dist.all_reduce(t)                        # the step's gradient collective: every rank must be inside
            steps.append(self.name)
            if self.fail_step_on == rank:
                return torch.tensor(float("nan"))
            return torch.tensor(1.0)

    def make(name, fail_build_on=None, fail_step_on=None):
        def build():
            built.append(name)
            if fail_build_on == rank:
                raise RuntimeError(f"capture of {name} failed on rank {rank}")
            return Step(name, fail_step_on)
        return (name, build)
    rungs = {"second": [make("seg", fail_build_on=1), make("single")],                         # rank 1 cannot capture the first mode
             "third": [make("seg", fail_build_on=0), make("single", fail_step_on=1)],          # ... and the second turns non-finite on rank 1
             "first": [make("seg"), make("single")]}[scenario]
    label, step = bench.choose_step(rungs, world, torch.device("cpu"), lambda: resets.append(1), trial_steps=2, log=lambda m: None)
    q.put((rank, label, list(built), list(steps), len(resets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario,want", [("first", "seg"), ("second", "single"), ("third", None)])
def test_launch_mode_ladder_two_ranks(scenario, want):
    """bench.choose_step: segmented graphs -> single graph -> eager.  A launch mode counts only if it captured AND survived its trial
    steps on EVERY rank (MIN all-reduce votes): when rank 1 cannot capture the first mode both ranks land on the second; when that one
    turns non-finite on one rank both fall through to eager (None) -- no rank is ever left alone in a collective, nothing hangs, and a
    rank that succeeded alone resets and follows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ladder_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, l0, b0, s0, n0), (r1, l1, b1, s1, n1) = res
    assert l0 == l1 == want, res
    assert b0 == b1, res                                            # both ranks tried the same rungs in the same order
    assert s0 == s1, res                                            # ... and ran the same trial steps (no rank stepped alone)
    if scenario == "first":
        assert b0 == ["seg"] and n0 == n1 == 0
    if scenario == "second":
        assert b0 == ["seg", "single"] and s0 == ["single", "single"] and n0 == n1 == 1
    if scenario == "third":
        assert n0 == n1 == 2 and s0 == ["single", "single"]


def _picks_worker(rank, world, port, q):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ocpg_amd.util import gemm_sync
    mine = torch.tensor([[11 + rank, 3], [-(1 << 62) + 5, 0], [77, 7]], dtype=torch.int64) if rank == 0 else torch.tensor([[99, 9]], dtype=torch.int64)
    got = []
    n = gemm_sync.share(torch.device("cpu"), src=0, export=lambda: mine, apply=lambda t: got.append(t.clone()))
    q.put((rank, n, [t.tolist() for t in got]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gemm_plan_choices_are_broadcast_from_rank0():
    """ocpg_amd.util.gemm_sync.share: rank 0's (plan key hash, candidate index) pairs reach every other rank unchanged (64-bit hashes
    included) and are applied there; rank 0 applies nothing; a rank's own timings never travel."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_picks_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == (0, 3, [])
    assert res[1] == (1, 3, [[[11, 3], [-(1 << 62) + 5, 0], [77, 7]]])
