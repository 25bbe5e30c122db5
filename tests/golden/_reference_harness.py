"""Import the reference (read-only /root/reference) on CPU with stand-ins for the packages this
image lacks.  Build-container only: /root/reference does not exist on the GPU box and nothing
under -m gpu / smoke() / bench.py imports this file.  See SURVEY.md section 8c + Appendix C.

Stand-ins installed (none of them is arithmetic on the hot path except (5), which calls the
reference's *own* pure-PyTorch sampler):
  (1) torchvision: version string, ops.boxes.box_area, models._utils.IntermediateLayerGetter
  (2) timm.models.layers: DropPath (identity in eval), trunc_normal_, to_2tuple
  (3) pycocotools.mask
  (4) an empty ``MultiScaleDeformableAttention`` module (CUDA-only extension)
  (5) MSDeformAttnFunction.apply -> reference ms_deform_attn_core_pytorch
  (6) RobertaModel/RobertaTokenizerFast.from_pretrained -> random-init roberta-base / fixed ids
"""
from __future__ import annotations

import argparse
import sys
import types

import torch
import transformers  # noqa: F401  (must be imported before the torchvision stand-in exists)
from transformers import RobertaConfig, RobertaModel, RobertaTokenizerFast

REFERENCE_ROOT = "/root/reference"


def roberta_base_config() -> RobertaConfig:
    return RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12,
                         num_attention_heads=12, intermediate_size=3072,
                         max_position_embeddings=514, type_vocab_size=1,
                         layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)


class FixedTokenizer:
    """Returns preset ids regardless of the strings (no vocab files offline)."""

    def __init__(self):
        self.ids = None
        self.attn = None        # optional attention mask (padded expressions in a batch)

    def batch_encode_plus(self, texts, padding="longest", return_tensors="pt"):
        from transformers import BatchEncoding
        ids = self.ids.clone()
        attn = torch.ones_like(ids) if self.attn is None else self.attn.clone()
        return BatchEncoding({"input_ids": ids, "attention_mask": attn})


def _install_stubs():
    if "torchvision" not in sys.modules or not hasattr(sys.modules["torchvision"], "_soc_stub"):
        tv = types.ModuleType("torchvision")
        tv.__version__ = "0.15.0"
        tv._soc_stub = True
        ops = types.ModuleType("torchvision.ops")
        boxes = types.ModuleType("torchvision.ops.boxes")
        boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        ops.boxes = boxes
        models = types.ModuleType("torchvision.models")
        _utils = types.ModuleType("torchvision.models._utils")
        _utils.IntermediateLayerGetter = type("IntermediateLayerGetter", (torch.nn.ModuleDict,), {})
        models._utils = _utils
        tv.ops, tv.models = ops, models
        sys.modules.update({"torchvision": tv, "torchvision.ops": ops,
                            "torchvision.ops.boxes": boxes, "torchvision.models": models,
                            "torchvision.models._utils": _utils})

    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert not self.training, "stand-in DropPath is eval-only"
            return x

    tl.DropPath = DropPath
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    tl.to_2tuple = lambda x: (x, x)
    timm.models, tm.layers = tm, tl
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})

    pc = types.ModuleType("pycocotools")
    pcm = types.ModuleType("pycocotools.mask")
    pc.mask = pcm
    sys.modules.update({"pycocotools": pc, "pycocotools.mask": pcm})

    sys.modules["MultiScaleDeformableAttention"] = types.ModuleType("MultiScaleDeformableAttention")


def import_reference():
    """Returns the reference's ``models`` package (build_model etc.)."""
    sys.dont_write_bytecode = True
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    RobertaModel.from_pretrained = classmethod(lambda cls, *a, **k: RobertaModel(roberta_base_config()))
    RobertaTokenizerFast.from_pretrained = classmethod(lambda cls, *a, **k: FixedTokenizer())

    import models.ops.functions.ms_deform_attn_func as ref_func
    import models.ops.modules.ms_deform_attn as ref_mod

    class _Shim:
        calls = []  # (value, shapes, lsi, loc, w, out) captured when recording is on
        record = False

        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            out = ref_func.ms_deform_attn_core_pytorch(value, shapes, loc, w)
            if _Shim.record:
                _Shim.calls.append(tuple(t.detach().clone() for t in (value, shapes, lsi, loc, w, out)))
            return out

    ref_mod.MSDeformAttnFunction = _Shim
    import models as ref_models
    ref_models._msda_shim = _Shim
    ref_models._msda_core = ref_func.ms_deform_attn_core_pytorch
    return ref_models


def reference_args(backbone: str = "video-swin-t") -> argparse.Namespace:
    """Values of configs/refer_youtube_vos.yaml (reference) that build_model reads."""
    return argparse.Namespace(
        backbone=backbone, backbone_pretrained_path=None, use_checkpoint=False,
        DeformTransformer=dict(enc_layers=3, dec_layers=3, dim_feedforward=2048, d_model=256,
                               dropout=0.1, nheads=8, num_queries=20, num_feature_levels=4,
                               dec_n_points=4, enc_n_points=4, two_stage=False),
        VOC=dict(input_dim=256, window_size=0, num_frame_queries=20, num_frames=8, num_queries=20,
                 nheads=8, dim_feedforward=2048, enc_layers=3, dec_layers=3),
        num_classes=1, rel_coord=True, with_box_refine=True,
        text_encoder_type="roberta-base", freeze_text_encoder=True,
        mask_kernels_dim=8, controller_layers=3, dynamic_mask_channels=8,
        vl_loss=True, aux_loss=True, device="cpu", dataset_name="ref_youtube_vos",
        set_cost_con=0, set_cost_cls=2, set_cost_dice=5, set_costs_box=2, set_costs_giou=2,
        class_loss_coef=2, con_loss_coef=1, sigmoid_focal_loss_coef=2, dice_loss_coef=5,
        eos_coef=0.1, box_loss_coef=2, giou_coef=2)
