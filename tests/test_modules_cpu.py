"""Product modules vs the reference's golden vectors on CPU (host logic; MSDA op replaced by the test double)."""
import pytest
import torch

import module_checks as mc

CPU = torch.device("cpu")


@pytest.fixture()
def msda_double(monkeypatch):
    from oracle.msda import MSDAOracleFunction
    import ocpg_amd.models.ops.modules.ms_deform_attn as mod
    monkeypatch.setattr(mod, "MSDeformAttnFunction", MSDAOracleFunction)


def test_lfm(golden):
    mc.check_lfm(golden("lfm"), CPU)


def test_fusion(golden):
    mc.check_fusion(golden("fusion"), CPU)


def test_msda_module(golden, msda_double):
    mc.check_msda_module(golden("msda_module"), CPU)


def test_transformer(golden, msda_double):
    mc.check_transformer(golden("transformer"), CPU)


def test_dynmask_mso(golden):
    mc.check_dynmask_mso(golden("dynmask_mso"), CPU)


def test_matcher_criterion(golden):
    mc.check_matcher_crit(golden("matcher_crit"), CPU)
