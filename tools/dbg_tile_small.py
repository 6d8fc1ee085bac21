import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from test_msda_gpu import _local_inputs
from ocpg_amd.models.ops.functions import ms_deform_attn_backward
from oracle import msda as om
dev = torch.device("cuda:0")
value, shapes, ls, loc, attn, go = _local_inputs(dev, 1, [(16, 24), (8, 12)])
ogv, ogl, oga = om.msda_c_backward(value, shapes, ls, loc, attn, go)
dv, dl, da, dg = (t.to(dev) for t in (value, loc, attn, go))
a, b_ = shapes.to(dev), shapes.to(dev)
a._ocpg_host = shapes
g1 = ms_deform_attn_backward(dv, a, ls.to(dev), dl, da, dg)
g2 = ms_deform_attn_backward(dv, b_, ls.to(dev), dl, da, dg)
os.environ["OCPG_MSDA_TILE"] = "0"
g3 = ms_deform_attn_backward(dv, a, ls.to(dev), dl, da, dg)
for name, x in (("tile", g1[0]), ("row", g2[0]), ("col", g3[0])):
    d = (x.cpu() - ogv).abs()
    i = d.argmax()
    print(name, "max abs err vs oracle", d.max().item(), "at", i.item(), "ref", ogv.flatten()[i].item(), "max ref", ogv.abs().max().item())
