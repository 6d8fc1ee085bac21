"""ocpg_bn_act_bwd (frozen-BN + ReLU backward of the ResNet body, channels-last bf16) at the step's shapes, HIP events around the C-ABI call,
warm and (BN_COLD=1) with 1 GiB written between calls.  One JSON line per shape."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd._lib import lib, stream_ptr

dev = torch.device("cuda:0")
cold_buf = torch.empty(1 << 28, dtype=torch.float32, device=dev) if os.environ.get("BN_COLD") == "1" else None
for name, rows, C, skip in (("layer2 conv3 (+skip)", 38400, 512, True), ("layer3 conv3 (+skip)", 9600, 1024, True), ("layer3 conv2", 9600, 256, False),
                            ("layer4 conv3 (+skip)", 2400, 2048, True)):
    g = torch.Generator().manual_seed(0)
    gy = torch.randn(rows, C, generator=g).to(dev, torch.bfloat16)
    y = torch.randn(rows, C, generator=g).to(dev, torch.bfloat16)
    scale = torch.rand(C, generator=g).to(dev) + 0.5
    gx, gs = torch.empty_like(gy), (torch.empty_like(gy) if skip else None)

    def call():
        rc = lib().ocpg_bn_act_bwd(gy.data_ptr(), y.data_ptr(), scale.data_ptr(), gx.data_ptr(), gs.data_ptr() if gs is not None else None,
                                   rows, C, 1, 1, 1, stream_ptr())
        assert rc == 0, rc
    call()
    ref = torch.where(y.float() > 0, gy.float(), torch.zeros(()).to(dev))
    assert torch.equal(gx, (ref * scale).to(torch.bfloat16)) and (gs is None or torch.equal(gs, ref.to(torch.bfloat16)))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        if cold_buf is not None:
            cold_buf.fill_(1.0)
        a.record(); call(); b.record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[15]
    byt = rows * C * 2 * (3 + (1 if skip else 0))
    print(json.dumps({"shape": name, "rows": rows, "C": C, "us": round(us, 1), "GBs": round(byt / us / 1e3, 1), "frac_of_8TBs": round(byt / us / 1e3 / 8000, 3)}), flush=True)
