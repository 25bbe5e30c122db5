"""Small library-level fusions around the GEMMs (fp32, results identical to the un-fused ops)."""
from __future__ import annotations

import torch
from torch import nn


def linear_relu(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    """relu(x @ W^T + b) with the ReLU applied in the GEMM epilogue (hipBLASLt RELU_BIAS): bit-identical
    to F.relu(lin(x)) and saves one read+write pass over the activation (316 MB per deformable
    encoder FFN at the BASELINE config)."""
    x2 = x.reshape(-1, x.shape[-1])
    y = torch._addmm_activation(lin.bias, x2, lin.weight.t(), use_gelu=False)
    return y.view(*x.shape[:-1], y.shape[-1])
