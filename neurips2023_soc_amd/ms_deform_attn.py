"""Multi-scale deformable attention: module + the reference's native-op seam, backed by K2.

`ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
im2col_step)` has the signature and error behaviour of the reference extension module
`MultiScaleDeformableAttention` (models/ops/src/vision.cpp:13-16, ms_deform_attn_cuda.cu:20-80);
`MSDeformAttn` mirrors models/ops/modules/ms_deform_attn.py:31-117 including the extra
(sampling_locations, attention_weights) return values the SOC fork added (:117).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from . import fused, hot_ops


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step: int):
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if step <= 0 or batch % step != 0:  # reference ms_deform_attn_cuda.cu:50-52
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    return hot_ops.msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step: int):
    """(grad_value, grad_sampling_loc, grad_attn_weight): the other entry point of the reference
    extension module (vision.cpp:13-16, ms_deform_attn_cuda.cu:83-153)."""
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if step <= 0 or batch % step != 0:
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    return hot_ops.msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)


class MSDeformAttnFunction(torch.autograd.Function):
    """The reference autograd Function (functions/ms_deform_attn_func.py:21-38) on K2 / K2 backward."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, w = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, w, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.zeros_(self.sampling_offsets.weight)
        ang = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        g = torch.stack([ang.cos(), ang.sin()], -1)
        g = (g / g.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        g = g * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(g.reshape(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None, pad_flag=None, return_sampling=True,
                query_pos=None, value=None, residual=None):
        """Reference signature plus optional arguments used by this package's own layers
        (``query_pos``: added to ``query`` inside the offset / weight projections; ``value``: the already
        projected ``value_proj(input_flatten)`` when the caller computed it ahead of time):
        ``residual``: added to the result in the output projection's epilogue (the encoder layer's shortcut);
        ``return_sampling=False`` (SOC never reads the sampling locations / weights) allows the fused
        kernel, which needs ``pad_flag`` = int32[1] device tensor "the padding mask has any True"."""
        N, Lq, _ = query.shape
        _, S, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        if value is None:
            value = fused.apply(self.value_proj, input_flatten)     # K13b on pixel-sized memories, K7 / library otherwise
        offsets_raw, logits_raw = fused.linear_multi(
            query, [(self.sampling_offsets.weight, self.sampling_offsets.bias, True),
                    (self.attention_weights.weight, self.attention_weights.bias, True)], query_pos)
        if (not return_sampling and value.is_cuda and L == 4 and P == 4 and self.d_model // M == 32
                and reference_points.shape[-1] in (2, 4)
                and (input_padding_mask is None or pad_flag is not None)):
            out = hot_ops.msda_fused_forward(
                value.view(N, S, M, 32), input_spatial_shapes, input_level_start_index, reference_points,
                offsets_raw.view(N, Lq, M, L, P, 2), logits_raw.view(N, Lq, M, L * P), input_padding_mask,
                pad_flag)
            return fused.linear(out, self.output_proj.weight, self.output_proj.bias, residual=residual), None, None
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, M, self.d_model // M)
        offsets = offsets_raw.view(N, Lq, M, L, P, 2)
        weights = F.softmax(logits_raw.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        if reference_points.shape[-1] == 2:
            normalizer = input_spatial_shapes.flip(-1).to(offsets.dtype)  # (W_l, H_l)
            loc = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.contiguous(), weights.contiguous(), self.im2col_step)
        out = self.output_proj(out)
        return (out if residual is None else out + residual), loc, weights
