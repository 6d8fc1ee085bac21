"""Reference-free descriptors/helpers shared by make_fixtures.py (build container) and the tests (anywhere).

Everything here only regenerates *inputs* from seeds; the expected outputs live in the *.npz files.
"""
import torch

import synth


def level_start(shapes):
    s = torch.as_tensor(shapes, dtype=torch.long)
    return s, torch.cat((s.new_zeros((1,)), s.prod(1).cumsum(0)[:-1]))



MSDA_CASES = [
    # name, N, M, D, shapes, Lq, P, loc range (lo, hi)
    dict(name="enc_like", N=2, M=8, D=32, shapes=[(8, 12), (4, 6), (2, 3), (1, 2)], Lq=None, P=4, lo=-0.15, hi=1.15),
    dict(name="dec_like", N=3, M=8, D=32, shapes=[(8, 12), (4, 6), (2, 3), (1, 2)], Lq=5, P=4, lo=0.0, hi=1.0),
    dict(name="odd_d", N=1, M=3, D=7, shapes=[(5, 7), (3, 2)], Lq=11, P=3, lo=-0.5, hi=1.5),
    dict(name="one_level", N=2, M=1, D=64, shapes=[(9, 5)], Lq=17, P=1, lo=-0.1, hi=1.1),
    dict(name="edges", N=1, M=2, D=4, shapes=[(4, 4), (2, 2)], Lq=8, P=2, lo=None, hi=None),
]


def msda_case_inputs(c):
    """Shared with the tests: regenerate a case's inputs from its descriptor."""
    shapes, lsi = level_start(c["shapes"])
    S = int(shapes.prod(1).sum())
    Lq = c["Lq"] or S
    L = len(c["shapes"])
    k = "msda_" + c["name"]
    value = synth.rand(k + "_v", (c["N"], S, c["M"], c["D"]))
    if c["lo"] is None:   # exact pixel centres / borders: h_im in {-1, -0.5, 0, H-1, H-0.5, H}
        grid = torch.tensor([-0.5, 0.0, 0.5, 1.0, 3.5, 4.0, 4.5, 2.0]) / 4.0
        loc = grid[torch.randint(0, 8, (c["N"], Lq, c["M"], L, c["P"], 2),
                                 generator=torch.Generator().manual_seed(7))]
    else:
        loc = synth.rand(k + "_l", (c["N"], Lq, c["M"], L, c["P"], 2), uniform=True) * (c["hi"] - c["lo"]) + c["lo"]
    attn = synth.rand(k + "_a", (c["N"], Lq, c["M"], L * c["P"]), uniform=True) + 1e-3
    attn = (attn / attn.sum(-1, keepdim=True)).view(c["N"], Lq, c["M"], L, c["P"])
    go = synth.rand(k + "_g", (c["N"], Lq, c["M"] * c["D"]))
    return value, shapes, lsi, loc, attn, go



def padded_masks(N, shapes, frac_h, frac_w):
    """Bool padding masks per level (True = padded), bottom/right padding like collate_fn produces."""
    out = []
    for (h, w) in shapes:
        m = torch.zeros(N, h, w, dtype=torch.bool)
        for n in range(N):
            vh = max(1, int(round(h * frac_h[n])))
            vw = max(1, int(round(w * frac_w[n])))
            m[n, vh:, :] = True
            m[n, :, vw:] = True
        out.append(m)
    return out



TINY = dict(backbone="resnet50", hidden_dim=64, mask_dim=64, dim_feedforward=128, enc_layers=1, dec_layers=2,
            num_frames=2, num_queries=3, num_feature_levels=4, dropout=0.0)


def tiny_text(B, L=7):
    tf = synth.rand("e2e_text", (B, L, 768))
    ts = synth.rand("e2e_sent", (B, 768))
    pm = torch.zeros(B, L, dtype=torch.bool)
    if B > 1:
        pm[1, 5:] = True
    return tf, ts, pm




from ocpg_amd.opts import default_args  # noqa: E402,F401  (single source: the package)


def e2e_inputs(B, T, H, W, sizes, device="cpu"):
    """The clip / mask / targets of the e2e fixture (same construction as make_fixtures.run_e2e)."""
    x = torch.zeros(B, T, 3, H, W)
    mask = torch.ones(B, T, H, W, dtype=torch.bool)
    targets = []
    for i, (h, w) in enumerate(sizes):
        x[i, :, :, :h, :w] = synth.rand(f"e2e_clip{i}", (T, 3, h, w))
        mask[i, :, :h, :w] = False
        targets.append(synth.synthetic_targets(1, T, h, w, device)[0])
    return x.to(device), mask.to(device), targets
