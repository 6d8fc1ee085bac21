import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocpg_amd.models.ops.functions.gemm_func import gemm
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.bfloat16):
    for (m, n, k) in ((1311, 32, 128), (1311, 128, 32), (1311, 128, 64), (1312, 32, 128), (9600, 256, 1024), (1311, 32, 32), (77, 40, 24)):
        for ta, tb in ((0, 1), (0, 0), (1, 0)):
            a = torch.randn((k, m) if ta else (m, k), device=dev, dtype=dt); b = torch.randn((n, k) if tb else (k, n), device=dev, dtype=dt)
            ref = ((a.t() if ta else a).float() @ (b.t() if tb else b).float())
            try:
                got = gemm(a, b, bool(ta), bool(tb)).float()
                err = (got - ref).abs().max().item() / ref.abs().max().item()
            except Exception as e:
                err = str(e)[:60]
            print(str(dt)[6:], (m, n, k), "ta%d tb%d" % (ta, tb), "rel err", err)
