"""Video-Swin restatement vs the reference's golden vectors (CPU; the e2e case swaps the MSDA op for the test double)."""
import pytest
import torch

import swin_checks as sc

CPU = torch.device("cpu")


@pytest.fixture()
def msda_double(monkeypatch):
    from oracle.msda import MSDAOracleFunction
    import ocpg_amd.models.ops.modules.ms_deform_attn as mod
    monkeypatch.setattr(mod, "MSDeformAttnFunction", MSDAOracleFunction)


def test_window_attention_and_masks(golden):
    sc.check_window_attention(golden("swin3d"), CPU)


def test_window_attention_full_window_n392(golden):
    sc.check_window_attention_n392(golden("swin_n392"), CPU)


def test_shifted_block_and_patch_merging(golden):
    sc.check_block_and_merging(golden("swin3d"), CPU)


def test_backbone_tiny(golden):
    sc.check_backbone(golden("swin3d"), CPU)


def test_e2e_with_video_swin(golden, msda_double):
    sc.check_e2e_swin(golden("e2e_swin"), CPU)
