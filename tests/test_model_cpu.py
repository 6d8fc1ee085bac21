"""Host logic of the product model on CPU against the reference's golden vectors.

The product's MSDeformAttn op has no CPU path (it raises); for these host-logic tests ONLY, the op is replaced by a
test double (the oracle's C restatement, itself pinned to the reference in test_oracle_msda.py).  The GPU suite
(test_model_gpu.py) runs the same checks with the real HIP op.
"""
import pytest
import torch

import model_checks


@pytest.fixture()
def msda_double(monkeypatch):
    from oracle.msda import MSDAOracleFunction
    import ocpg_amd.models.ops.modules.ms_deform_attn as mod
    monkeypatch.setattr(mod, "MSDeformAttnFunction", MSDAOracleFunction)


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_train_step_matches_reference(golden, msda_double, tag):
    res = model_checks.run_train_step(golden("e2e_tiny"), tag, torch.device("cpu"), rtol=2e-4, atol=2e-5)
    print(res)


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_eval_tail_matches_reference(golden, msda_double, tag):
    model_checks.run_eval(golden("e2e_tiny"), tag, torch.device("cpu"), rtol=2e-4, atol=2e-5)
