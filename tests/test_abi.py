"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(ocpg_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for s in ("ocpg_msda_fwd_f32", "ocpg_msda_bwd_f32", "ocpg_msda_fwd_f64", "ocpg_msda_bwd_f64"):
        assert s in syms


def test_library_loads_and_exports_every_declared_symbol():
    from ocpg_amd import _lib
    from ocpg_amd.csrc import build
    build.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} declared in include/ but not exported by libocpg_hip.so"
    # the ctypes table binds exactly the declared compute entry points
    assert set(_lib.SIGNATURES) == set(declared_symbols()) - {"ocpg_hip_version"}
    assert _lib.lib().ocpg_hip_version().startswith(b"ocpg_hip gfx950")


def test_product_op_refuses_cpu_tensors():
    """Same contract as ms_deform_attn.h:38 (AT_ERROR 'Not implemented on the CPU'): no silent CPU fallback."""
    from ocpg_amd.models.ops.functions import MSDeformAttnFunction
    v = torch.zeros(1, 4, 1, 4)
    shapes = torch.tensor([[2, 2]])
    ls = torch.tensor([0])
    loc = torch.zeros(1, 1, 1, 1, 1, 2)
    attn = torch.ones(1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDeformAttnFunction.apply(v, shapes, ls, loc, attn, 64)


def test_product_never_imports_oracle():
    for path in glob.glob(os.path.join(ROOT, "ocpg_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "from oracle" not in src and "import oracle" not in src, path
