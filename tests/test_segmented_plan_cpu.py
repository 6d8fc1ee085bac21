"""bench.py's SegmentedGraphStep off the GPU: the PLAN (parameter groups, cut activations, hops) on the real backbones' parameter
names, and the capture-independent three-part backward (`segmented_backward`) against a plain backward on stand-in models that have
the backbones' wiring (every stage output feeds the neck AND the next stage): main.py:62's bucket-by-bucket overlap needs each part's
gradients to be exactly the whole-graph gradients."""
import contextlib
import copy
import os
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _Wrap(nn.Module):
    def __init__(self, body):
        super().__init__()
        self.body = body

    def forward(self, x):
        return self.body(x)


class _Stage(nn.Sequential):
    def __init__(self, c, depth):
        super().__init__(*[nn.Sequential(nn.Linear(c, c), nn.Tanh()) for _ in range(depth)])


class _SwinLikeBody(nn.Module):
    """patch_embed -> stage 0 -> [merge -> stage i]: `layers`, `downsamples` (last None), `patch_embed`, as VideoSwinTransformerBackbone"""

    def __init__(self, n_stages):
        super().__init__()
        self.patch_embed = nn.Linear(6, 8)
        self.layers = nn.ModuleList([_Stage(8 * 2 ** i, 2) for i in range(n_stages)])
        self.downsamples = nn.ModuleList([nn.Linear(8 * 2 ** i, 16 * 2 ** i) for i in range(n_stages - 1)] + [None])
        self.downsamples[-1] = None

    def forward(self, x):
        x = self.patch_embed(x)
        out = []
        for layer, down in zip(self.layers, self.downsamples):
            x = layer(x)
            out.append(x)
            if down is not None:
                x = down(x)
        return out


class _ResNetLikeBody(nn.Module):
    def __init__(self, blocks3):
        super().__init__()
        self.layer1 = _Stage(8, 1)
        self.layer2 = nn.Sequential(nn.Linear(8, 12), _Stage(12, 2))
        self.layer3 = nn.Sequential(*[nn.Sequential(nn.Linear(12, 12), nn.Tanh()) for _ in range(blocks3)])
        self.layer4 = nn.Sequential(nn.Linear(12, 16), _Stage(16, 1))
        for p in self.layer1.parameters():
            p.requires_grad_(False)            # the frozen stem / layer1 of the reference's ResNet

    def forward(self, x):
        f4 = self.layer1(x)
        f8 = self.layer2(f4)
        f16 = self.layer3(f8)
        f32 = self.layer4(f16)
        return [f4, f8, f16, f32]


class _Model(nn.Module):
    """neck: one Linear per backbone output (all of them are consumed, like features[:2] by the mask head and the rest by input_proj)"""

    def __init__(self, body, widths):
        super().__init__()
        self.backbone = nn.Sequential(_Wrap(body))
        self.neck = nn.ModuleList([nn.Linear(w, 4) for w in widths])
        self.head = nn.Linear(4, 1)

    def forward(self, x):
        feats = self.backbone[0](x)
        return self.head(sum(torch.tanh(n(f)) for n, f in zip(self.neck, feats))).square().mean()


def _check(model, x):
    twin = copy.deepcopy(model)
    twin(x).backward()
    want = {n: p.grad for n, p in twin.named_parameters() if p.requires_grad}
    group_of, cuts, hops = bench.SegmentedGraphStep.plan(model)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    P = [[p for n, p in named if group_of(n) == i] for i in range(5)]
    entered = []

    @contextlib.contextmanager
    def part(i):
        entered.append(i)
        yield
    lists, groups, _ = bench.SegmentedGraphStep.segmented_backward(lambda: model(x), P, cuts, hops, part, last=lambda: entered.append("last"))
    assert entered == [0, 1, 2, "last"]
    assert groups[0] == [0] and sorted(k for g in groups for k in g) == [k for k in range(5) if P[k]]
    names = {id(p): n for n, p in named}
    seen = 0
    for gl, gr in zip(lists, groups):
        ps = [p for k in gr for p in P[k]]
        assert len(ps) == len(gl)
        for p, g in zip(ps, gl):
            assert g is not None, names[id(p)]
            torch.testing.assert_close(g, want[names[id(p)]], rtol=1e-5, atol=1e-7)
            seen += 1
    assert seen == len(named)
    # the hooks are gone: a plain forward is the uncut model again
    model.zero_grad()
    model(x).backward()
    for n, p in named:
        torch.testing.assert_close(p.grad, want[n], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n_stages", [4, 3])
def test_segmented_backward_swin_wiring(n_stages):
    torch.manual_seed(0)
    model = _Model(_SwinLikeBody(n_stages), [8 * 2 ** i for i in range(n_stages)])
    assert bench.SegmentedGraphStep._kind(model) == "swin"
    assert bench.SegmentedGraphStep.supported(model, torch.float16) and bench.SegmentedGraphStep.supported(model, torch.bfloat16)
    assert not bench.SegmentedGraphStep.supported(model, None)
    _check(model, torch.randn(5, 6))


@pytest.mark.parametrize("blocks3", [23, 2])
def test_segmented_backward_resnet_wiring(blocks3):
    torch.manual_seed(1)
    model = _Model(_ResNetLikeBody(blocks3), [8, 12, 12, 16])
    assert bench.SegmentedGraphStep._kind(model) == "resnet"
    _check(model, torch.randn(5, 8))


@pytest.mark.parametrize("levels", [4, 3])
def test_plan_covers_every_video_swin_parameter(levels):
    """On the real Video-Swin backbone's parameter names: every stage, patch merging and the patch embedding falls into the group of a
    hop, the two LAST stages into the second graph (bucket 1), the earlier ones into the third."""
    from ocpg_amd.models.video_swin_transformer import Backbone

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = nn.Sequential(Backbone("video_swin_t_p4w7", output_levels=levels))
            self.rest = nn.Linear(2, 2)
    m = M()
    n_st = len(m.backbone[0].body.layers)
    assert n_st == levels
    group_of, cuts, hops = bench.SegmentedGraphStep.plan(m)
    assert [k for k, _ in cuts] == [f"s{i}" for i in range(n_st)]
    hop_groups = [g for hs in hops for _, g, _ in hs]
    assert len(set(hop_groups)) == len(hop_groups)
    by = {}
    for n, p in m.named_parameters():
        by.setdefault(group_of(n), []).append(n)
    assert set(by) == {0} | set(hop_groups)
    assert all(n.startswith("rest.") for n in by[0])
    last = [n for n in by[hops[0][0][1]]]
    assert all(f".layers.{n_st - 1}." in n or f".downsamples.{n_st - 2}." in n for n in last) and last
    first_group = hops[1][-1][1]
    assert any(".patch_embed." in n for n in by[first_group]) and all(".patch_embed." in n or ".layers.0." in n for n in by[first_group])
    # chained: each hop starts where the previous one delivered
    flat = [h for hs in hops for h in hs]
    assert [h[2] for h in flat[:-1]] == [h[0] for h in flat[1:]] and flat[-1][2] is None and flat[0][0] == f"s{n_st - 1}"
