"""FPN spatial decoder feeding the dynamic mask head (reference models/segmentation.py:11-74).
conv3x3 + GroupNorm(8) + ReLU ladder with nearest up-sampling; library convolutions (MIOpen).  On the GPU the
convolutions run without their bias and everything between two convolutions is one launch: K17 (bias + GroupNorm +
ReLU) and K18 (adapter bias + nearest up-sampling + add).

`forward_tokens` is the form SOC.forward_tail uses on the GPU: the encoder memory and the stride-4 backbone map are
token-major already, so the whole ladder runs channels-last -- K19 (implicit-GEMM 3x3 convolution on MFMA, reading the
memory levels in place), K10 (GroupNorm + ReLU on tokens), the 1x1 adapters as GEMMs over tokens, K18 (token form) -- and
only the last convolution writes NCHW, which is what the dynamic mask head (K4) reads."""
from __future__ import annotations

from typing import List

import torch.nn.functional as F
import torch
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

    def _taps(self, conv: nn.Conv2d) -> Tensor:
        """conv.weight [Cout, Cin, 3, 3] as K19's [Cout, 9*Cin], cached per weight version"""
        cache = self.__dict__.setdefault("_tap_cache", {})
        key = id(conv)
        w = conv.weight
        ent = cache.get(key)
        if ent is None or ent[0] != (w.data_ptr(), w._version, str(w.device)):
            cache[key] = ((w.data_ptr(), w._version, str(w.device)),
                          w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())
        return cache[key][1]

    def tokens_supported(self, memory: Tensor) -> bool:
        convs = [self.lay1, self.lay2, self.lay3, self.lay4] + ([self.lay5] if self.add_extra_layer else []) + [self.out_lay]
        gns = [self.gn1, self.gn2, self.gn3, self.gn4] + ([self.gn5] if self.add_extra_layer else [])
        return (memory.is_cuda and memory.dtype == torch.float32 and not self.training
                and all(c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.in_channels % 16 == 0
                        for c in convs)
                and all(hot_ops.groupnorm_tokens_supported(g.num_channels, g.num_groups) for g in gns))

    def forward_tokens(self, memory: Tensor, shapes, feats0: Tensor) -> Tensor:
        """memory [n, S, C]: the deformable encoder's output, levels concatenated finest first; shapes: their (h, w);
        feats0 [n, C0, H/4, W/4] channels-last in memory (the backbone's stride-4 map) -> [n, mask_kernels_dim, H/4, W/4].
        The same ladder as forward(memory_maps[-1], [memory_maps[1], memory_maps[0], feats0])."""
        n = memory.shape[0]
        starts, at = [], 0
        for (h, w) in shapes:
            starts.append(at)
            at += h * w

        def level(l):
            h, w = shapes[l]
            return memory[:, starts[l]:starts[l] + h * w], (h, w)

        def conv_gn_relu(x, hw, lay, gn):
            y = hot_ops.conv3x3_tokens(x, hw, self._taps(lay), lay.bias)
            return hot_ops.groupnorm_tokens(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu=True)

        x, hw = level(2)
        x = conv_gn_relu(x, hw, self.lay1, self.gn1)
        x = conv_gn_relu(x, hw, self.lay2, self.gn2)
        laterals = [level(1), level(0)]
        stages = [(self.adapter1, self.lay3, self.gn3), (self.adapter2, self.lay4, self.gn4)]
        if self.add_extra_layer:
            h0, w0 = feats0.shape[-2:]
            tok0 = feats0.permute(0, 2, 3, 1)
            if not tok0.is_contiguous():
                tok0 = tok0.contiguous()
            laterals.append((tok0.view(n, h0 * w0, -1), (h0, w0)))
            stages.append((self.adapter3, self.lay5, self.gn5))
        for (feat, fhw), (adapter, lay, gn) in zip(laterals, stages):
            wa = adapter.weight.view(adapter.out_channels, -1)                   # the 1x1 convolution as a GEMM over tokens
            if feat.is_contiguous() and hot_ops.ws_linear_supported(feat.view(-1, feat.shape[-1]), wa, False):
                lateral = hot_ops.ws_linear(feat.view(-1, feat.shape[-1]), wa, None).view(n, -1, wa.shape[0])
            else:       # a level slice of the memory: batched over the frames, no copy of the strided input
                lateral = torch.bmm(feat, wa.t().unsqueeze(0).expand(n, -1, -1))
            x = hot_ops.upsample_add_tokens(lateral, adapter.bias, x, fhw, hw)
            hw = fhw
            x = conv_gn_relu(x, hw, lay, gn)
        return hot_ops.conv3x3_tokens(x, hw, self._taps(self.out_lay), self.out_lay.bias, out_nchw=True)

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
