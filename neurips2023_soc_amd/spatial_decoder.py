"""FPN spatial decoder feeding the dynamic mask head (reference models/segmentation.py:11-74).
conv3x3 + GroupNorm(8) + ReLU ladder with nearest up-sampling; library convolutions (MIOpen).  On the GPU the
convolutions run without their bias and everything between two convolutions is one launch: K17 (bias + GroupNorm +
ReLU) and K18 (adapter bias + nearest up-sampling + add)."""
from __future__ import annotations

from typing import List

import torch.nn.functional as F
from torch import Tensor, nn

from . import hot_ops


class FPNSpatialDecoder(nn.Module):
    def __init__(self, context_dim: int, fpn_dims: List[int], mask_kernels_dim: int = 8):
        super().__init__()
        d = [context_dim, context_dim // 2, context_dim // 4, context_dim // 8, context_dim // 16]
        self.lay1, self.gn1 = nn.Conv2d(context_dim, d[0], 3, padding=1), nn.GroupNorm(8, d[0])
        self.lay2, self.gn2 = nn.Conv2d(d[0], d[1], 3, padding=1), nn.GroupNorm(8, d[1])
        self.lay3, self.gn3 = nn.Conv2d(d[1], d[2], 3, padding=1), nn.GroupNorm(8, d[2])
        self.lay4, self.gn4 = nn.Conv2d(d[2], d[3], 3, padding=1), nn.GroupNorm(8, d[3])
        self.adapter1 = nn.Conv2d(fpn_dims[0], d[1], 1)
        self.adapter2 = nn.Conv2d(fpn_dims[1], d[2], 1)
        self.context_dim = context_dim
        self.add_extra_layer = len(fpn_dims) == 3
        if self.add_extra_layer:
            self.adapter3 = nn.Conv2d(fpn_dims[2], d[3], 1)
            self.lay5, self.gn5 = nn.Conv2d(d[3], d[4], 3, padding=1), nn.GroupNorm(8, d[4])
            self.out_lay = nn.Conv2d(d[4], mask_kernels_dim, 3, padding=1)
        else:
            self.out_lay = nn.Conv2d(d[3], mask_kernels_dim, 3, padding=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.zeros_(m.bias)

    @staticmethod
    def _conv_gn_relu(x: Tensor, lay: nn.Conv2d, gn: nn.GroupNorm) -> Tensor:
        y = F.conv2d(x, lay.weight, None, lay.stride, lay.padding)
        if hot_ops.groupnorm_nchw_supported(y, gn.num_groups):
            return hot_ops.groupnorm_nchw(y, lay.bias, gn.weight, gn.bias, gn.num_groups, gn.eps, relu=True)
        return F.relu(gn(y + lay.bias.view(1, -1, 1, 1)))

    def forward(self, x: Tensor, layer_features: List[Tensor]) -> Tensor:
        stages = [(self.adapter1, self.lay3, self.gn3), (self.adapter2, self.lay4, self.gn4)]
        if self.add_extra_layer:
            stages.append((self.adapter3, self.lay5, self.gn5))
        if not x.is_cuda:
            x = F.relu(self.gn1(self.lay1(x)))
            x = F.relu(self.gn2(self.lay2(x)))
            for feat, (adapter, lay, gn) in zip(layer_features, stages):
                lateral = adapter(feat)
                x = lateral + F.interpolate(x, size=lateral.shape[-2:], mode="nearest")
                x = F.relu(gn(lay(x)))
            return self.out_lay(x)
        x = self._conv_gn_relu(x, self.lay1, self.gn1)
        x = self._conv_gn_relu(x, self.lay2, self.gn2)
        for feat, (adapter, lay, gn) in zip(layer_features, stages):
            lateral = F.conv2d(feat, adapter.weight, None)
            x = hot_ops.upsample_add_nchw(lateral, adapter.bias, x)
            x = self._conv_gn_relu(x, lay, gn)
        return self.out_lay(x)
