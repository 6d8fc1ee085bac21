"""Argument defaults of the hot path -- host-side mirror of the reference's opts.get_args_parser() (opts.py:3-156) for
the fields build_model / the criterion / the optimizer groups read.  bench.py, the tests and the fixture generator all
build their Namespace here, so the product's driver never imports from the test tree."""


def default_args(**over):
    """The reference's opts.get_args_parser() defaults (opts.py:3-156) for the fields the model reads, plus the
    flags every launch script sets (`--with_box_refine --binary --freeze_text_encoder`, main.py:33-34 masks/binary)."""
    import argparse
    ns = argparse.Namespace(
        lr=1e-4, lr_backbone=5e-5, lr_text_encoder=1e-5, lr_linear_proj_mult=1.0, weight_decay=5e-4, clip_max_norm=0.1,
        lr_backbone_names=["backbone.0"], lr_text_encoder_names=["text_encoder"],
        lr_linear_proj_names=["reference_points", "sampling_offsets"], amp=False,
        masks=True, binary=True, with_box_refine=True, freeze_text_encoder=True, two_stage=False,
        device="cpu", dataset_file="ytvos", backbone="resnet50", text_backbone="Roberta", backbone_pretrained=None,
        use_checkpoint=False, dilation=False, position_embedding="sine", num_feature_levels=4, output_levels=4,
        enc_layers=4, dec_layers=4, dim_feedforward=2048, hidden_dim=256, dropout=0.1, nheads=8, num_frames=3,
        num_queries=5, dec_n_points=4, enc_n_points=4, pre_norm=False, freeze_video_encoder=False, mask_dim=256,
        controller_layers=2, dynamic_mask_channels=16, rel_coord=True, aux_loss=True,
        set_cost_class=2, set_cost_bbox=5, set_cost_giou=2, set_cost_mask=2, set_cost_boundary=2, set_cost_dice=5,
        mask_loss_coef=2, boundary_loss_coef=2, dice_loss_coef=5, proj_loss_coef=5, lst_loss_coef=2, cls_loss_coef=2,
        bbox_loss_coef=5, giou_loss_coef=2, eos_coef=0.1, focal_alpha=0.25, eval=False, seed=42,
        text_encoder_lazy=True)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns
