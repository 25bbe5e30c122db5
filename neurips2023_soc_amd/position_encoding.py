"""Sine positional encodings (reference models/position_encoding.py:11-82)."""
from __future__ import annotations

import math

import torch
from torch import nn

from .nested_tensor import NestedTensor


def _dim_t(n: int, temperature: float, device) -> torch.Tensor:
    i = torch.arange(n, dtype=torch.float32, device=device)
    return temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / n)


def _interleave_sin_cos(p: torch.Tensor) -> torch.Tensor:
    return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)


class PositionEmbeddingSine1D(nn.Module):
    """Token positions of a padded text batch: NestedTensor([B,C,L] or any, mask [B,L]) -> [B,C,L]."""

    def __init__(self, num_pos_feats=256, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor) -> torch.Tensor:
        mask = tensor_list.mask
        assert mask is not None
        x = (~mask).cumsum(1, dtype=torch.float32)
        if self.normalize:
            x = x / (x[:, -1:] + 1e-6) * self.scale
        p = _interleave_sin_cos(x[:, :, None] / _dim_t(self.num_pos_feats, self.temperature, mask.device))
        return p.permute(0, 2, 1)


class PositionEmbeddingSine2D(nn.Module):
    """Pixel positions: mask [N,H,W] -> [N, 2*num_pos_feats, H, W] (y half first, then x half)."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor) -> torch.Tensor:
        mask = tensor_list.mask
        assert mask is not None
        keep = ~mask
        y = keep.cumsum(1, dtype=torch.float32)
        x = keep.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y = (y - 0.5) / (y[:, -1:, :] + 1e-6) * self.scale
            x = (x - 0.5) / (x[:, :, -1:] + 1e-6) * self.scale
        d = _dim_t(self.num_pos_feats, self.temperature, mask.device)
        px = _interleave_sin_cos(x[..., None] / d)
        py = _interleave_sin_cos(y[..., None] / d)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)
