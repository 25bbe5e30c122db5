"""Video-Swin backbone on MI355X (reference models/video_swin_transformer.py).

MI355X-first data flow per block: LayerNorm -> ONE qkv GEMM over the un-padded token-major map
-> HIP kernel K1 (soc_win_attn3d_f32) which does pad / cyclic roll / window partition / relative
position bias / shift mask / softmax / PV / window reverse / un-roll / crop in a single launch
-> proj GEMM -> residual -> LayerNorm -> MLP.  The reference materialises padded, rolled and
partitioned copies plus a [nW*nH, 392, 392] score tensor per block (551 MB at stage 0); none of
that exists here.  Module / parameter names follow the reference so its checkpoints load
(`backbone.0.body.{patch_embed,layers.N.blocks.M.*,downsamples.N.*}`).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from . import fused, hot_ops
from .nested_tensor import NestedTensor
from .position_encoding import PositionEmbeddingSine2D

SWIN_CONFIGS = {  # reference :733-779
    "video-swin-t": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24)),
    "video-swin-s": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24)),
    "video-swin-b": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
}
WINDOW = (8, 7, 7)
PATCH = (1, 4, 4)


def relative_position_index(window: Sequence[int]) -> torch.Tensor:
    """Buffer kept for checkpoint compatibility (reference :113-128); K1 derives it implicitly."""
    axes = [torch.arange(w) for w in window]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij")).flatten(1)
    rel = (grid[:, :, None] - grid[:, None, :]).permute(1, 2, 0).contiguous()
    for a in range(3):
        rel[:, :, a] += window[a] - 1
    rel[:, :, 0] *= (2 * window[1] - 1) * (2 * window[2] - 1)
    rel[:, :, 1] *= 2 * window[2] - 1
    return rel.sum(-1)


class WindowAttention3D(nn.Module):
    """Parameter holder + launcher of K1.  forward takes the *token-major* normed map."""

    def __init__(self, dim: int, window_size: Sequence[int], num_heads: int):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        n_rel = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) * (2 * window_size[2] - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(n_rel, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x: torch.Tensor, shift: Sequence[int], residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B,D,H,W,C] (after norm1, un-padded) -> attention branch output [B,D,H,W,C] (+ residual: the block's
        shortcut, added in the projection's epilogue where a hand-written kernel runs it: fused.route_linear)."""
        qkv = fused.linear(x, self.qkv.weight, self.qkv.bias)           # stage 2: K24 (K13b with SOC_SPLIT_OFF=k24)
        attn = hot_ops.window_attention3d(qkv, self.qkv.bias, self.relative_position_bias_table,
                                          self.num_heads, self.window_size, shift)
        return fused.linear(attn, self.proj.weight, self.proj.bias, residual=residual)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(fused.linear_gelu(x, self.fc1))


class SwinTransformerBlock3D(nn.Module):
    def __init__(self, dim: int, num_heads: int, window_size, shift_size, mlp_ratio: float = 4.0):
        super().__init__()
        self.shift_size = tuple(shift_size)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention3D(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Un-fused form (used only for a stand-alone block); BasicLayer runs the fused schedule."""
        x = x + self.attn(self.norm1(x), self.shift_size)
        return x + self.mlp(self.norm2(x))


class BasicLayer(nn.Module):
    """One stage; ``downsample`` lives outside (reference VideoSwinTransformerBackbone :663-670)."""

    def __init__(self, dim: int, depth: int, num_heads: int, window_size):
        super().__init__()
        shift = tuple(w // 2 for w in window_size)
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim, num_heads, window_size, (0, 0, 0) if i % 2 == 0 else shift)
            for i in range(depth)])
        self.downsample = None
        self.out_tag = None     # "swin<stage>": the stage's last K23 launch writes into hot_ops.placed(out_tag) when one is set

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # [B,D,H,W,C] token-major throughout
        """Tall stages with a short channel width (stages 0-1 of every Video-Swin variant) run each block as four
        K13 launches around K1 -- norm1 + qkv | K1 | proj + residual | norm2 + fc1 + GELU | fc2 + residual -- so no
        LayerNorm-ed or GELU-ed copy of the token map is ever a separate pass.  Elsewhere every residual add is fused
        with the LayerNorm that consumes its result (K5): x += attn -> norm2, x += mlp -> next block's norm1.  Same
        values as block-by-block."""
        blocks = self.blocks
        flow = self.stage_flow(x)
        if flow == "k20":
            # K20 (bf16 matrix cores, exact three-way split): LayerNorm folded into the layer, GELU / residual in its
            # epilogue -- norm1 + qkv | K1 | proj + residual | norm2 + fc1 + GELU | fc2 + residual
            for blk in blocks:
                a, m = blk.attn, blk.mlp
                qkv = hot_ops.linear_split(x, a.qkv.weight, a.qkv.bias, ln=(blk.norm1.weight, blk.norm1.bias, blk.norm1.eps))
                o = hot_ops.window_attention3d(qkv, a.qkv.bias, a.relative_position_bias_table, a.num_heads,
                                               a.window_size, blk.shift_size)
                x = hot_ops.linear_split(o, a.proj.weight, a.proj.bias, residual=x)
                if fused.mlp_ok(x, m.fc1, m.fc2):      # C = 256 (Swin-B stage 1): norm2 + fc1 + GELU + fc2 + shortcut in one K23 launch
                    x = hot_ops.mlp_split(x, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, "gelu",
                                          ln=(blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), residual=x,
                                          out=hot_ops.placed(self.out_tag, x) if blk is blocks[-1] else None)
                    continue
                h = hot_ops.linear_split(x, m.fc1.weight, m.fc1.bias, ln=(blk.norm2.weight, blk.norm2.bias, blk.norm2.eps),
                                         act="gelu")
                if _FORCE_SPLIT_FLOW:
                    x = hot_ops.linear_split(h, m.fc2.weight, m.fc2.bias, residual=x)
                else:
                    x = fused.linear(h, m.fc2.weight, m.fc2.bias, residual=x)  # K20 or library + add (split_wins)
            return x
        if flow == "ws":
            for blk in blocks:
                a = blk.attn
                qkv = hot_ops.ws_linear(x, a.qkv.weight, a.qkv.bias, ln=(blk.norm1.weight, blk.norm1.bias, blk.norm1.eps))
                o = hot_ops.window_attention3d(qkv, a.qkv.bias, a.relative_position_bias_table, a.num_heads,
                                               a.window_size, blk.shift_size)
                x = hot_ops.ws_linear(o, a.proj.weight, a.proj.bias, residual=x)
                if fused.mlp_ok(x, blk.mlp.fc1, blk.mlp.fc2):
                    # K23: norm2 + fc1 + GELU + fc2 + residual in one launch, the hidden layer never leaves the registers
                    x = hot_ops.mlp_split(x, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias, "gelu",
                                          ln=(blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), residual=x,
                                          out=hot_ops.placed(self.out_tag, x) if blk is blocks[-1] else None)
                    continue
                h = hot_ops.ws_linear(x, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                                      ln=(blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), act="gelu")
                if hot_ops.ws_linear_supported(h, blk.mlp.fc2.weight, False):
                    x = hot_ops.ws_linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, residual=x)
                else:
                    x = x + blk.mlp.fc2(h)
            return x
        if flow == "k24":
            for blk in blocks:
                a, m = blk.attn, blk.mlp
                qkv = hot_ops.xs_linear(x, a.qkv.weight, a.qkv.bias, ln=(blk.norm1.weight, blk.norm1.bias, blk.norm1.eps))
                o = hot_ops.window_attention3d(qkv, a.qkv.bias, a.relative_position_bias_table, a.num_heads,
                                               a.window_size, blk.shift_size)
                x = hot_ops.xs_linear(o, a.proj.weight, a.proj.bias, residual=x)
                h = hot_ops.xs_linear(x, m.fc1.weight, m.fc1.bias, ln=(blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), act="gelu")
                x = fused.linear(h, m.fc2.weight, m.fc2.bias, residual=x)
            return x
        _, h = hot_ops.add_layernorm(x, None, blocks[0].norm1.weight, blocks[0].norm1.bias, blocks[0].norm1.eps)
        if flow == "k23":
            # stage 2 (C = 384): the shortcut rides in the projection's epilogue, norm2 + fc1 + GELU + fc2 + shortcut are one K23
            # launch that also emits norm1 of the next block: no LayerNorm pass inside the stage
            for i, blk in enumerate(blocks):
                x = blk.attn(h, blk.shift_size, residual=x)
                m = blk.mlp
                nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else None
                y = hot_ops.mlp_split(x, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, "gelu",
                                      ln=(blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), residual=x,
                                      post_ln=None if nxt is None else (nxt.weight, nxt.bias, nxt.eps), return_sum=nxt is not None)
                x, h = y if nxt is not None else (y, None)        # norm1 of the next block leaves in the same launch
            return x
        for i, blk in enumerate(blocks):
            a = blk.attn(h, blk.shift_size)
            x, h = hot_ops.add_layernorm(x, a, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            m = blk.mlp(h)
            if i + 1 < len(blocks):
                nxt = blocks[i + 1].norm1
                x, h = hot_ops.add_layernorm(x, m, nxt.weight, nxt.bias, nxt.eps)
            else:
                x = x + m
        return x

    def stage_flow(self, x) -> str:
        """How the blocks of this stage run (shapes, dtypes and devices only -- tests/test_routes.py pins it per configuration):
        "ws"  K13b launches around K1, LayerNorms in their prologues, the MLP as one K23 launch where K23 covers the width
              (tall stages of width 96 / 128 / 192: stages 0-1 of Swin-T / -S, stage 0 of Swin-B);
        "k20" every layer on K20, LayerNorm folded into the packed weights (tall stages wider than that: Swin-B stage 1, and
              stage 1 of the others with SOC_SPLIT_OFF=k13);
        "k23" one LayerNorm pass at the stage's entry, then per block qkv | K1 | proj + shortcut | K23 (norm2 + MLP + shortcut +
              norm1 of the next block): stage 2 of Swin-T / -S (C = 384) and of Swin-B (C = 512);
        "k24" norm1 + qkv | K1 | proj + shortcut | norm2 + fc1 + GELU as K24 launches (LayerNorm in their prologues), fc2 + shortcut
              through fused.linear: stage 3 (C = 768, Swin-B 1024; the hidden width 4 C is beyond K24);
        "k5"  every shortcut add fused with the LayerNorm that consumes it (K5), GEMMs per fused.route_*: the rest."""
        blk = self.blocks[0]
        # stage 1 (C = 192) in the split arithmetic: K13b beats K20 on qkv / proj / fc1 (49 / 24 / 71 against 60 / 30 / 74 us,
        # tools/experiments/k13b_time.py); with SOC_SPLIT_OFF=k13 the K20 flow takes it
        prefer_ws = hot_ops.k13_split_enabled() and x.shape[-1] in hot_ops.WS_SPLIT_LN_K and self._weight_stationary(x)
        # C = 384 / 512 keep the "k23" flow however many rows a launch group brings (four clips: 29 440): K24 qkv 161 us against
        # K20 + LayerNorm 229, K23 with norm1 of the next block 392 against 415 + a LayerNorm in the next qkv
        # (tools/experiments/stage2_group_flow.py); the "k20" flow is for the tall stages K24 / K23 do not cover that way
        prefer_k23 = (x.shape[-1] in (384, 512) and fused.mlp_ok(x, blk.mlp.fc1, blk.mlp.fc2)
                      and all(fused.xs_ok(x, w) for w in (blk.attn.qkv.weight, blk.attn.proj.weight)))
        if not prefer_ws and not prefer_k23 and self._split_flow(x):
            return "k20"
        if self._weight_stationary(x):
            return "ws"
        if fused.mlp_ok(x, blk.mlp.fc1, blk.mlp.fc2):
            return "k23"
        if (x.shape[-1] in (768, 1024) and x.numel() // x.shape[-1] >= 1024
                and all(hot_ops.xs_linear_supported(x, w) for w in (blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc1.weight))):
            return "k24"
        return "k5"

    def _split_flow(self, x: torch.Tensor) -> bool:
        """Tall stages wider than K13's sweet spot (C >= 192: stage 1 of Swin-T / -S, stages 1 of Swin-B): every layer on
        K20 (tools/split_probe.py: 66 / 29 / 80 us against 75 / 36 / 99-108 us for K13 at Swin-T stage 1)."""
        blk = self.blocks[0]
        C = x.shape[-1]
        rows = x.numel() // C
        if _FORCE_SPLIT_FLOW:       # experiment switch (tools/experiments/README.md): every stage with rows >= the value
            return (x.is_cuda and x.dtype == torch.float32 and C >= 192 and rows >= _FORCE_SPLIT_FLOW
                    and hot_ops.split_enabled()
                    and hot_ops.linear_split_supported(x, blk.attn.qkv.weight, True))
        return (x.is_cuda and x.dtype == torch.float32 and C >= 192 and rows >= 16384
                and hot_ops.split_wins(rows, 3 * C, C, 1, "swin") and hot_ops.split_wins(rows, C, C, 1, "swin")
                and hot_ops.linear_split_supported(x, blk.attn.qkv.weight, True))

    def _weight_stationary(self, x: torch.Tensor) -> bool:
        """K13 pays where the token map is tall and the weights of a layer fit a CU's LDS in a few column ranges."""
        blk = self.blocks[0]
        return (x.is_cuda and x.numel() // x.shape[-1] >= 16384
                and hot_ops.ws_linear_supported(x, blk.attn.qkv.weight, True)
                and hot_ops.ws_linear_supported(x, blk.attn.proj.weight, False)
                and hot_ops.ws_linear_supported(x, blk.mlp.fc1.weight, True))


_FORCE_SPLIT_FLOW = int(os.environ.get("SOC_FORCE_SPLIT_FLOW_ROWS", "0"))


class PatchMerging(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = hot_ops.patch_merge_layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return fused.linear(h, self.reduction.weight, None)      # K13b (stage 0 -> 1), K24 (1 -> 2), library (2 -> 3: K = 1536)

    def forward_unfused(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's sequence of ops (pad, four strided slices, cat, norm, reduction)."""
        H, W = x.shape[2], x.shape[3]
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
        _, h = hot_ops.add_layernorm(x, None, self.norm.weight, self.norm.bias, self.norm.eps)
        return self.reduction(h)


class PatchEmbed3D(nn.Module):
    def __init__(self, embed_dim: int):
        super().__init__()
        self.proj = nn.Conv3d(3, embed_dim, kernel_size=PATCH, stride=PATCH)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B,3,T,H,W] -> token-major [B,T,H/4,W/4,C]; the (1,4,4) conv is a per-frame 4x4/4 conv."""
        B, _, T, H, W = x.shape
        frames = x.transpose(1, 2)                                      # [B,T,3,H,W]: the caller's own layout
        if frames.is_contiguous() and hot_ops.patch_embed_supported(frames.reshape(B * T, 3, H, W), self.proj.weight):
            # K21: convolution + LayerNorm in one pass over the clip (edge padding included)
            f = hot_ops.patch_embed_layernorm(frames.reshape(B * T, 3, H, W), self.proj.weight, self.proj.bias,
                                              self.norm.weight, self.norm.bias, self.norm.eps)
            return f.view(B, T, f.shape[1], f.shape[2], -1)
        if W % 4:
            x = F.pad(x, (0, 4 - W % 4))
        if H % 4:
            x = F.pad(x, (0, 0, 0, 4 - H % 4))
        Hp, Wp = x.shape[-2] // 4, x.shape[-1] // 4
        if x.is_cuda:
            # non-overlapping 4x4 patches: the convolution is a GEMM over (c, kh, kw) = 48 inputs per token,
            # whose output is already token-major (no NCHW -> NHWC copy of the 44 MB map)
            cols = x.transpose(1, 2).reshape(B * T, 3, Hp, 4, Wp, 4).permute(0, 2, 4, 1, 3, 5)
            f = F.linear(cols.reshape(B * T * Hp * Wp, 48), self.proj.weight.view(-1, 48), self.proj.bias)
            f = f.view(B, T, Hp, Wp, -1)
        else:
            f = x.transpose(1, 2).reshape(B * T, 3, x.shape[-2], x.shape[-1])
            f = F.conv2d(f, self.proj.weight[:, :, 0], self.proj.bias, stride=4)
            f = f.permute(0, 2, 3, 1).reshape(B, T, Hp, Wp, -1)
        return hot_ops.add_layernorm(f, None, self.norm.weight, self.norm.bias, self.norm.eps)[1]


class VideoSwinTransformerBackbone(nn.Module):
    def __init__(self, name: str):
        super().__init__()
        cfg = SWIN_CONFIGS[name]
        d = cfg["embed_dim"]
        self.patch_embed = PatchEmbed3D(d)
        self.layers = nn.ModuleList([BasicLayer(d * 2 ** i, cfg["depths"][i], cfg["num_heads"][i], WINDOW)
                                     for i in range(4)])
        for i, layer in enumerate(self.layers):
            layer.out_tag = f"swin{i}"
        self.downsamples = nn.ModuleList([PatchMerging(d * 2 ** i) for i in range(3)] + [None])
        self.layer_output_channels = [d * 2 ** i for i in range(4)]

    def forward(self, samples: torch.Tensor, num_frames: int) -> Dict[str, torch.Tensor]:
        n, c, h, w = samples.shape
        x = samples.view(n // num_frames, num_frames, c, h, w).transpose(1, 2)
        x = self.patch_embed(x)
        out = {}
        for i, layer in enumerate(self.layers):
            x = layer(x)
            B, T, hh, ww, C = x.shape
            out[str(i)] = x.permute(0, 1, 4, 2, 3).reshape(B * T, C, hh, ww)  # '(b t) c h w'
            if self.downsamples[i] is not None:
                x = self.downsamples[i](x)
        return out


_NO_PAD: Dict = {}


def resize_pad_mask(mask: torch.Tensor, size, unpadded: bool = False) -> torch.Tensor:
    """Padding mask [N,H,W] at a feature level's size (nearest, reference :716-722).  ``unpadded``: the caller vouches
    that the mask is all-False, so the result depends on the geometry alone and is a cached constant (three launches
    per level otherwise, on the critical path between the backbone and the fusion)."""
    if unpadded:
        key = (mask.shape[0], tuple(size), str(mask.device))
        if key not in _NO_PAD:
            if len(_NO_PAD) > 64:
                _NO_PAD.clear()
            _NO_PAD[key] = torch.zeros(mask.shape[0], *size, dtype=torch.bool, device=mask.device)
        return _NO_PAD[key]
    return F.interpolate(mask[None].float(), size=size).to(torch.bool)[0]


class Backbone(nn.Module):
    """reference BackboneBase/Backbone :701-730"""

    def __init__(self, name: str):
        super().__init__()
        self.body = VideoSwinTransformerBackbone(name)
        self.strides = [4, 8, 16, 32]
        self.num_channels = list(self.body.layer_output_channels)

    def forward(self, tensor_list: NestedTensor, num_frames: int) -> Dict[str, NestedTensor]:
        xs = self.body(tensor_list.tensors, num_frames)
        unpadded = bool(getattr(tensor_list, "unpadded", False))
        return {k: NestedTensor(x, resize_pad_mask(tensor_list.mask, x.shape[-2:], unpadded)) for k, x in xs.items()}


class Joiner(nn.Sequential):
    """backbone[0] = Backbone, backbone[1] = position embedding (reference :781-800).

    Like the reference it rewrites ``samples`` in place to the '(b t)' layout."""

    def __init__(self, backbone: Backbone, position_embedding: nn.Module):
        super().__init__(backbone, position_embedding)
        self.strides, self.num_channels = backbone.strides, backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        t, b = tensor_list.tensors.shape[:2]
        tensor_list.tensors = tensor_list.tensors.transpose(0, 1).flatten(0, 1)
        tensor_list.mask = tensor_list.mask.transpose(0, 1).flatten(0, 1)
        xs = self[0](tensor_list, num_frames=t)
        out: List[NestedTensor] = [xs[k] for k in sorted(xs)]
        # The stride-4 level's encoding ([T,256,H/4,W/4] = 118 MB at 360x640) is never read by
        # SOC.forward (it uses pos[-3:], reference models/soc.py:226); only the last three are built.
        # For un-padded batches (every single-video batch) the encodings depend on the geometry
        # alone and are cached, like the reference caches compute_mask with lru_cache.
        unpadded = bool(getattr(tensor_list, "unpadded", False))
        pos: List[Optional[torch.Tensor]] = [None]
        for x in out[1:]:
            pos.append(self.position_encoding(x, unpadded))
        return out, pos

    def position_encoding(self, x: NestedTensor, unpadded: bool = False) -> torch.Tensor:
        if not unpadded:
            return self[1](x).to(x.tensors.dtype)
        cache = self.__dict__.setdefault("_pos_cache", {})
        key = (tuple(x.mask.shape), str(x.mask.device))
        if key not in cache:
            cache[key] = self[1](NestedTensor(x.tensors, torch.zeros_like(x.mask))).to(x.tensors.dtype)
        return cache[key]


def build_video_swin_backbone(args) -> Joiner:
    pe = PositionEmbeddingSine2D(args.DeformTransformer["d_model"] // 2, normalize=True)
    model = Joiner(Backbone(args.backbone), pe)
    path = getattr(args, "backbone_pretrained_path", None)
    if isinstance(path, str):
        load_kinetics_weights(model[0].body, path)
    return model


def load_kinetics_weights(body: VideoSwinTransformerBackbone, path: str) -> None:
    """Kinetics-400 Video-Swin checkpoint -> backbone (reference :651-661): strip 'backbone.',
    sum the patch-embed kernel over its temporal taps, move layers.N.downsample.* to downsamples.N.*"""
    sd = torch.load(path, map_location="cpu")["state_dict"]
    sd = {k[9:]: v for k, v in sd.items() if k.startswith("backbone.")}
    sd["patch_embed.proj.weight"] = sd["patch_embed.proj.weight"].sum(dim=2, keepdim=True)
    remap = {}
    for k, v in sd.items():
        parts = k.split(".")
        if len(parts) > 3 and parts[0] == "layers" and parts[2] == "downsample":
            k = ".".join(["downsamples", parts[1]] + parts[3:])
        if k.startswith("norm."):
            continue
        remap[k] = v
    # the reference loads this checkpoint strictly (video_swin_transformer.py:660); the only tolerated differences
    # here are the buffers this implementation does not keep (relative_position_index, cached attn masks)
    missing, unexpected = body.load_state_dict(remap, strict=False)
    derived = ("relative_position_index", "attn_mask")
    missing = [k for k in missing if not k.endswith(derived)]
    unexpected = [k for k in unexpected if not k.endswith(derived)]
    if missing or unexpected:
        raise RuntimeError(f"Kinetics checkpoint {path} does not match the backbone: missing {missing[:8]} "
                           f"unexpected {unexpected[:8]}")
