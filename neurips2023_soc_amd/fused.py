"""Linear layers of the hot path: which GEMM runs them (fp32 throughout).

* many rows (pixels / tokens): the library GEMM (hipBLASLt / rocBLAS through torch, kernel chosen by
  gemm_tuning's table), with the ReLU in its epilogue where one follows;
* few rows (frame queries, video queries, words): K7 `soc_linear_small_f32`, which also folds in the
  positional add in front of the layer and the ReLU behind it.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import hot_ops


def is_small(x: torch.Tensor) -> bool:
    """True when linear(x, ...) runs in K7 rather than in the library GEMM."""
    K = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and K % 16 == 0
            and x.numel() // K <= hot_ops.SMALL_LINEAR_MAX_ROWS)


def ffn_relu(x: torch.Tensor, lin1: nn.Linear, lin2: nn.Linear) -> torch.Tensor:
    """lin2(relu(lin1(x))) -- the deformable encoder's feed-forward block (reference models/deformable_transformer.py:
    253-263): K23 in one launch where it covers the shape (the encoder layer calls it with the residual and norm2 folded in as
    well), the two-GEMM path otherwise (SOC_MATMUL=f32, SOC_SPLIT_OFF=mlp, few rows)."""
    if mlp_ok(x, lin1, lin2):
        return hot_ops.mlp_split(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, "relu")
    return apply(lin2, linear_relu(x, lin1))


def mlp_ok(x: torch.Tensor, lin1: nn.Linear, lin2: nn.Linear) -> bool:
    """K23 takes the whole two-layer block lin2(act(lin1(LN(x)))) (+ residual, + LayerNorm behind it): pixel-sized inputs of
    width 96 / 128 / 192 / 256 / 384 / 512 (Video-Swin stages 0-2, the deformable encoder's feed-forward block)."""
    return (lin1.bias is not None and lin2.bias is not None and x.numel() // x.shape[-1] >= 4096
            and hot_ops.mlp_split_supported(x, lin1.weight, lin2.weight))


def ws_dense_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """K13b's step-by-step form (K = 384 / 512, no LayerNorm in front) for a plain linear layer on a pixel-sized, already
    normalised input: Video-Swin stage 2 qkv / proj / fc1 (53 / 26 / 71 us against the library's 74 / 30 and K20's 75-80,
    tools/experiments/k13b_time.py)."""
    K = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and K in (384, 512) and hot_ops.k13_split_enabled()
            and x.numel() // K >= 4096 and hot_ops.ws_linear_supported(x, weight, False))


def ws_plain_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """K13b for a linear layer without a LayerNorm in front: the K = 384 / 512 form above, or the K = 96 / 128 / 192 form on
    a tall input (patch-merging reductions, input_proj of the finer levels)."""
    K = x.shape[-1]
    return ws_dense_ok(x, weight) or (
        x.is_cuda and x.dtype == torch.float32 and K in (96, 128, 192, 256) and hot_ops.k13_split_enabled()
        and x.numel() // K >= 16384 and hot_ops.ws_linear_supported(x, weight, False))


def w256_tall_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """256 x 256 layers at the row counts of a launch group (the encoder's value_proj / output_proj + shortcut over 385 600 rows, the
    fusion blocks' projections over 288 000): K20's tiles edge out K13b's four column ranges and K24's streamed weights there
    (round 6, tools/experiments/w256_probe.py: 335 / 378 / 346 us at 385 600 rows, 242 / 265 / 259 at 288 000; K13b keeps 73 600)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and tuple(weight.shape) == (256, 256)
            and x.numel() // 256 >= 200_000 and hot_ops.split_enabled() and not _SPLIT_OFF_SITES()
            and hot_ops.linear_split_supported(x, weight))


def _SPLIT_OFF_SITES() -> bool:
    from .matmul_mode import _SPLIT_OFF
    return "plain" in _SPLIT_OFF or "w256" in _SPLIT_OFF


def _split_ok(x: torch.Tensor, weight: torch.Tensor, fused_passes: int, site: str = "plain") -> bool:
    K = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and hot_ops.linear_split_supported(x, weight)
            and hot_ops.split_wins(x.numel() // K, weight.shape[0], K, fused_passes, site))


def xs_ok(x, weight) -> bool:
    """K24 (x split once per row tile, weights streamed) where it measured faster than K13b / K20 / the library
    (tools/experiments/k24_time.py): input widths 384 / 768 (512 / 1024 for Swin-B) on pixel-sized inputs (Video-Swin stage 2 / 3 qkv, proj, fc1, the
    patch-merging reduction into stage 2), and width 256 up to 40 960 rows (the fusion blocks' query projection, the encoder's
    value_proj / output_proj at 38 560 rows: 41 us against K13b's 47, tools/experiments/k24_main_tail.py)."""
    K = x.shape[-1]
    rows = x.numel() // K
    return (rows * weight.shape[0] >= 2_000_000 and (K in (384, 512, 768, 1024) or (K == 256 and 16384 <= rows <= 40960))
            and hot_ops.xs_linear_supported(x, weight))       # (narrow outputs on few rows: the library's small tiles win)


def route_linear(x, weight, bias=None, add=None, relu: bool = False, mul=None, residual=None) -> str:
    """Which kernel runs mul * act((x [+ add]) @ weight.T + bias) + residual: "k7" (few rows), "k24", "k13b", "k20", or "library".
    Decided from shapes, dtypes and devices only (tests/test_routes.py drives it with stand-ins for every layer of the five
    BASELINE configurations and pins the answers, so that a silent fall-back to the library fails a test)."""
    if is_small(x):
        return "k7"
    if add is None and mul is None and w256_tall_ok(x, weight):
        return "k20"
    if add is None and mul is None and xs_ok(x, weight):
        return "k24"
    if add is None and mul is None and ws_plain_ok(x, weight):
        return "k13b"
    n_fused = int(add is not None) + int(relu) + int(mul is not None) + int(residual is not None)
    site = "relu" if relu else ("mul" if mul is not None else ("res" if residual is not None else ("add" if add is not None else "plain")))
    if _split_ok(x, weight, n_fused, site) and (add is None or tuple(add.shape) == tuple(x.shape)):
        return "k20"
    return "library"


def route_gelu(x, weight) -> str:
    """Which kernel runs gelu(x @ weight.T + bias): "k24", "k13b", "k20", "k12" or "library" (+ a GELU pass)."""
    K = x.shape[-1]
    rows = x.numel() // K
    if xs_ok(x, weight):
        return "k24"
    if ws_dense_ok(x, weight):
        return "k13b"
    if _split_ok(x, weight, 1, "gelu"):
        return "k20"
    if x.is_cuda and x.dtype == torch.float32 and rows >= 16384 and K <= 256 and K % 16 == 0 and weight.shape[0] % 4 == 0:
        return "k12"
    return "library"


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           add: Optional[torch.Tensor] = None, relu: bool = False, mul: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mul * act((x [+ add]) @ weight.T + bias) + residual; `add` broadcasts like `x + add`."""
    K = x.shape[-1]
    route = route_linear(x, weight, bias, add, relu, mul, residual)
    if route == "k7":
        if add is not None and add.shape != x.shape:
            add = add.expand_as(x)
        y = hot_ops.linear_small(x, weight, bias, add, relu)
        if mul is not None:
            y = y * mul
        return y if residual is None else y + residual
    if route == "k24":
        return hot_ops.xs_linear(x, weight, bias, None, residual, "relu" if relu else "none")
    if route == "k13b":
        return hot_ops.ws_linear(x, weight, bias, None, residual, "relu" if relu else "none")
    if route == "k20":
        # the positional add in front and ReLU / mul / residual behind are part of the launch
        return hot_ops.linear_split(x, weight, bias, None, residual, "relu" if relu else "none", add, mul)
    if mul is not None or residual is not None:
        y = linear(x, weight, bias, add, relu)
        if mul is not None:
            y = y * mul
        return y if residual is None else y + residual
    if add is not None:
        x = x + add
    if relu and bias is not None:
        x2 = x.reshape(-1, K)
        y = torch._addmm_activation(bias, x2, weight.t(), use_gelu=False)
        return y.view(*x.shape[:-1], y.shape[-1])
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


def route_linear_multi(x, layers, add=None) -> str:
    """Which kernel runs several layers on the same (x [+ add]): "k7" (few rows, up to four layers), "k20" (two stacked layers
    as one two-output launch, x + pos in the loader), "k12" (the f32-MFMA form of the same) or "library"."""
    if is_small(x) and len(layers) <= 4:
        return "k7"
    K = x.shape[-1]
    two_on_sum = (x.is_cuda and x.dtype == torch.float32 and len(layers) == 2 and add is not None
                  and tuple(add.shape) == tuple(x.shape) and all(u for _, _, u in layers)
                  and all(w.shape[0] % 4 == 0 for w, _, _ in layers))
    if two_on_sum and all(b is not None for _, b, _ in layers) and _split_ok(x, layers[0][0], 2, "multi"):
        return "k20"
    if two_on_sum and K % 16 == 0 and K <= 256:
        return "k12"
    return "library"


def linear_multi(x: torch.Tensor, layers, add: Optional[torch.Tensor] = None):
    """[(x + add if use_add else x) @ W.T + b for (W, b, use_add) in layers] on the kernel route_linear_multi names."""
    route = route_linear_multi(x, layers, add)
    if route == "k7":
        return hot_ops.linear_small_multi(x, layers, add)
    if route == "k20":
        # K20 with two outputs: both layers' rows stacked into one weight image (cached), x + pos in the loader
        w, b = _stacked(layers[0][0], layers[0][1], layers[1][0], layers[1][1])
        return list(hot_ops.linear_split(x, w, b, add=add, split_at=layers[0][0].shape[0]))
    if route == "k12":
        # two pixel-sized layers on x + pos (the deformable encoder's sampling offsets and attention weights):
        # one K12 launch with the add in its prologue instead of an add kernel and two library GEMMs
        return hot_ops.linear_act_multi(x, [(w, b) for w, b, _ in layers], add)
    xa = x + add if (add is not None and any(u for _, _, u in layers)) else x
    return [F.linear(xa if u else x, w, b) for w, b, u in layers]


_stack_cache = hot_ops.DerivedCache()


def _stacked(w0, b0, w1, b1):
    """cat of two layers' (weight, bias), cached until one of them changes (K20 packs the stacked matrix once) and freed
    with the layers."""
    return _stack_cache.get((w0, b0, w1, b1), lambda: (torch.cat([w0.detach(), w1.detach()], 0).contiguous(),
                                                      torch.cat([b0.detach(), b1.detach()], 0).contiguous()))


def linear_relu(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    """relu(x @ W^T + b).  Many rows: the ReLU runs in the GEMM epilogue (hipBLASLt RELU_BIAS),
    bit-identical to F.relu(lin(x)) and one read+write pass over the activation less (316 MB per
    deformable-encoder FFN at the BASELINE config).  Few rows: K7."""
    return linear(x, lin.weight, lin.bias, relu=True)


def linear_gelu(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    """gelu(lin(x)), exact (erf) GELU, on the kernel route_gelu names: K13b / K20 apply it to the accumulators; K12 (tall inputs
    with a short reduction, f32 MFMA) likewise; elsewhere the library GEMM + a GELU pass."""
    route = route_gelu(x, lin.weight)
    if route == "k24":
        return hot_ops.xs_linear(x, lin.weight, lin.bias, act="gelu")
    if route == "k13b":
        return hot_ops.ws_linear(x, lin.weight, lin.bias, act="gelu")
    if route == "k20":
        return hot_ops.linear_split(x, lin.weight, lin.bias, act="gelu")
    if route == "k12":
        return hot_ops.linear_act(x, lin.weight, lin.bias, "gelu")
    return F.gelu(lin(x))


def apply(lin: nn.Linear, x: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """lin(x [+ add]) through `linear`."""
    return linear(x, lin.weight, lin.bias, add=add)
