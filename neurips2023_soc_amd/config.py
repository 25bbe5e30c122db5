"""Model hyper-parameters of the shipped configs (reference configs/refer_youtube_vos.yaml:17-118;
all six YAML files agree on the model section) and the reference's YAML flattening rule."""
from __future__ import annotations

import argparse


def default_args(backbone: str = "video-swin-t", **overrides) -> argparse.Namespace:
    ns = argparse.Namespace(
        backbone=backbone, backbone_pretrained_path=None, use_checkpoint=False,
        DeformTransformer=dict(enc_layers=3, dec_layers=3, dim_feedforward=2048, d_model=256, dropout=0.1,
                               nheads=8, num_queries=20, num_feature_levels=4, dec_n_points=4,
                               enc_n_points=4, two_stage=False),
        VOC=dict(input_dim=256, window_size=0, num_frame_queries=20, num_frames=8, num_queries=20,
                 nheads=8, dim_feedforward=2048, enc_layers=3, dec_layers=3),
        num_classes=1, rel_coord=True, with_box_refine=True, text_encoder_type="roberta-base",
        freeze_text_encoder=True, text_encoder_random_init=False, mask_kernels_dim=8,
        controller_layers=3, dynamic_mask_channels=8, vl_loss=True, aux_loss=True, device="cuda",
        dataset_name="ref_youtube_vos")
    for k, v in overrides.items():
        setattr(ns, k, v)
    return ns


def flatten_yaml_config(cfg: dict, cli: dict | None = None) -> argparse.Namespace:
    """{key: {desc?, value}} -> {key: value}, CLI wins (reference infer_refytb.py:351-355)."""
    flat = {k: (v["value"] if isinstance(v, dict) and "value" in v else v) for k, v in cfg.items()}
    flat.update({k: v for k, v in (cli or {}).items() if v is not None})
    return argparse.Namespace(**flat)
