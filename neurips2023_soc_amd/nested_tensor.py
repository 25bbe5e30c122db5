"""Padded-batch container and helpers on the model boundary.

Mirrors the reference's data-interchange types so callers (infer_refytb.py:206-214 style code)
work unchanged: NestedTensor (misc.py:103-122), nested_tensor_from_videos_list (misc.py:143-160),
inverse_sigmoid (misc.py:427-431).
"""
from __future__ import annotations

from typing import List, Optional

import torch

Tensor = torch.Tensor


class NestedTensor:
    """``tensors`` plus a bool ``mask`` that is True on padding."""

    def __init__(self, tensors: Tensor, mask: Optional[Tensor], unpadded: bool = False):
        self.tensors = tensors
        self.mask = mask
        # host-side knowledge "mask is all False" (set by nested_tensor_from_videos_list for batches of
        # equal-sized videos): lets the model reuse geometry-only constants without a device sync
        self.unpadded = unpadded

    def to(self, device) -> "NestedTensor":
        m = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), m, self.unpadded)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self) -> str:
        return repr(self.tensors)


def nested_tensor_from_videos_list(videos: List[Tensor]) -> NestedTensor:
    """list of [T,C,H,W] -> tensors [T,B,C,Hmax,Wmax] (zero padded) + mask [T,B,Hmax,Wmax]."""
    dims = [max(v.shape[i] for v in videos) for i in range(4)]
    t, c, h, w = dims
    b = len(videos)
    out = torch.zeros((b, t, c, h, w), dtype=videos[0].dtype, device=videos[0].device)
    mask = torch.ones((b, t, h, w), dtype=torch.bool, device=videos[0].device)
    for i, v in enumerate(videos):
        out[i, :v.shape[0], :, :v.shape[2], :v.shape[3]] = v
        mask[i, :v.shape[0], :v.shape[2], :v.shape[3]] = False
    same = all(tuple(v.shape) == tuple(videos[0].shape) for v in videos)
    return NestedTensor(out.transpose(0, 1), mask.transpose(0, 1), unpadded=same)


def inverse_sigmoid(x: Tensor, eps: float = 1e-5) -> Tensor:
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
