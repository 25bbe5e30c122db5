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


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           add: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """act((x [+ add]) @ weight.T + bias); `add` broadcasts like `x + add`."""
    K = x.shape[-1]
    if is_small(x):
        if add is not None and add.shape != x.shape:
            add = add.expand_as(x)
        return hot_ops.linear_small(x, weight, bias, add, relu)
    if add is not None:
        x = x + add
    if relu and bias is not None:
        x2 = x.reshape(-1, K)
        y = torch._addmm_activation(bias, x2, weight.t(), use_gelu=False)
        return y.view(*x.shape[:-1], y.shape[-1])
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


def linear_multi(x: torch.Tensor, layers, add: Optional[torch.Tensor] = None):
    """[(x + add if use_add else x) @ W.T + b for (W, b, use_add) in layers]: one K7 launch for few rows,
    one library GEMM per layer otherwise."""
    if is_small(x) and len(layers) <= 4:
        return hot_ops.linear_small_multi(x, layers, add)
    K = x.shape[-1]
    if (x.is_cuda and x.dtype == torch.float32 and len(layers) == 2 and add is not None and add.shape == x.shape
            and all(u for _, _, u in layers) and K % 16 == 0 and K <= 256
            and all(w.shape[0] % 4 == 0 for w, _, _ in layers)):
        # two pixel-sized layers on x + pos (the deformable encoder's sampling offsets and attention weights):
        # one K12 launch with the add in its prologue instead of an add kernel and two library GEMMs
        return hot_ops.linear_act_multi(x, [(w, b) for w, b, _ in layers], add)
    xa = x + add if (add is not None and any(u for _, _, u in layers)) else x
    return [F.linear(xa if u else x, w, b) for w, b, u in layers]


def linear_relu(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    """relu(x @ W^T + b).  Many rows: the ReLU runs in the GEMM epilogue (hipBLASLt RELU_BIAS),
    bit-identical to F.relu(lin(x)) and one read+write pass over the activation less (316 MB per
    deformable-encoder FFN at the BASELINE config).  Few rows: K7."""
    return linear(x, lin.weight, lin.bias, relu=True)


def linear_gelu(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    """gelu(lin(x)), exact (erf) GELU.  Tall inputs with a short reduction (Video-Swin stages 0-1: >= 16384
    tokens, K <= 256) run as K12 with the GELU applied to the accumulators; elsewhere the library GEMM is
    faster than K12 by more than the separate GELU pass costs (tools/gemm_probe.py) and is kept."""
    K = x.shape[-1]
    rows = x.numel() // K
    if (x.is_cuda and x.dtype == torch.float32 and rows >= 16384 and K <= 256 and K % 16 == 0
            and lin.out_features % 4 == 0):
        return hot_ops.linear_act(x, lin.weight, lin.bias, "gelu")
    return F.gelu(lin(x))


def apply(lin: nn.Linear, x: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """lin(x [+ add]) through `linear`."""
    return linear(x, lin.weight, lin.bias, add=add)
