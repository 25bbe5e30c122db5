"""Result selection / mask post-processing of the inference drivers.

select_trajectory + upsample_and_threshold restate infer_refytb.py:216-231;
ReferYoutubeVOSPostProcess mirrors models/postprocessing.py:193-234; merge_davis_objects the
multi-object argmax of infer_davis.py:264-272.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn


def select_trajectory(outputs: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """B=1: (query index [], mask logits [T,h,w]) of the query with the best mean sigmoid score."""
    scores = outputs["pred_cls"][:, 0].sigmoid().mean(0).max(-1)[0]
    idx = scores.argmax(-1)
    return idx, outputs["pred_masks"][:, 0].index_select(1, idx.view(1))[:, 0]


def upsample_and_threshold(mask_logits: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """[T,h,w] logits -> bool [T,H0,W0] (bilinear, align_corners=False, sigmoid>0.5).
    On the GPU this is the fused HIP kernel K6; CPU tensors (host-side post-processing of gathered
    results) take the two torch ops the reference uses."""
    if mask_logits.is_cuda:
        from . import hot_ops
        return hot_ops.upsample_threshold(mask_logits, size)
    up = F.interpolate(mask_logits[None], size=tuple(size), mode="bilinear", align_corners=False)[0]
    return up.sigmoid() > 0.5


def merge_davis_objects(soft_masks: torch.Tensor, threshold: float = 0.5, background: float = 0.1) -> torch.Tensor:
    """soft_masks [O,T,H,W] sigmoid scores of O objects -> label map [T,H,W] (0 = background)."""
    m = torch.where(soft_masks < threshold, torch.zeros_like(soft_masks), soft_masks)
    bg = torch.full_like(m[:1], background)
    return torch.cat([bg, m], 0).argmax(0)


class ReferYoutubeVOSPostProcess(nn.Module):
    @torch.inference_mode()
    def forward(self, outputs, videos_metadata, samples_shape_with_padding):
        prob = outputs["pred_cls"].sigmoid().mean(0)          # [b, nq, k]
        best = prob.max(-1)[0].argmax(-1)                      # [b]
        masks = outputs["pred_masks"].permute(1, 0, 2, 3, 4)   # b t nq h w
        b = masks.shape[0]
        masks = masks[torch.arange(b, device=masks.device), :, best]
        masks = F.interpolate(masks, size=samples_shape_with_padding, mode="bilinear", align_corners=False)
        masks = masks.sigmoid() > 0.5
        preds = []
        for vm, meta in zip(masks, videos_metadata):
            rh, rw = meta["resized_frame_size"]
            vm = vm[:, :rh, :rw].unsqueeze(1)
            vm = F.interpolate(vm.float(), size=meta["original_frame_size"], mode="nearest")
            preds.append({**meta, "pred_masks": vm.to(torch.uint8).cpu()})
        return preds


def build_postprocessors(dataset_name: str):
    """reference models/soc.py:648-660; only the Ref-YouTube-VOS / DAVIS inference paths are built."""
    if dataset_name in ("ref_youtube_vos", "joint"):
        return ReferYoutubeVOSPostProcess()
    if dataset_name == "davis":
        return None
    raise NotImplementedError(f"post-processor for {dataset_name!r} is outside the inference hot path "
                              "(A2D/JHMDB/COCO evaluation: SURVEY.md section 2, out of scope)")
