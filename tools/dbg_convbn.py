import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocpg_amd.models import amp_cache, backbone
dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.bfloat16):
  for project in (False, True):
    torch.manual_seed(3)
    blk = backbone.Bottleneck(64 if project else 128, 32, 1, 1, project).to(dev)
    for m in blk.modules():
        if isinstance(m, backbone.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.1), m.running_mean.normal_(0, 0.1), m.running_var.uniform_(0.5, 1.5)
        if isinstance(m, torch.nn.Conv2d):
            m.to(memory_format=torch.channels_last)
    blk = blk.to(dtype)
    x = torch.randn(3, 64 if project else 128, 19, 23, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    go = torch.randn(3, 128, 19, 23, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    res = []
    for mode in ("fused", "gemm", "miopen"):
        backbone.FUSED_CONV_BN = mode == "fused"; amp_cache.GEMM_1X1 = mode != "miopen"
        xi = x.clone().requires_grad_(True); blk.zero_grad()
        y = blk(xi); y.backward(go)
        res.append([("y", y.detach().float()), ("gx", xi.grad.float())] + [(n, p.grad.float()) for n, p in blk.named_parameters()])
    backbone.FUSED_CONV_BN = True; amp_cache.GEMM_1X1 = True
    for i, name in ((0, "fused"), (1, "gemm")):
        print(str(dtype)[6:], "project", project, name, "vs miopen:", " ".join("%s %.3g/%.3g" % (n, (a - b).abs().max().item(), b.abs().max().item()) for (n, a), (_, b) in zip(res[i], res[2])))
