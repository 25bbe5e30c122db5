"""Multi-head attention module whose core runs in the HIP kernel K3 (soc_xattn_f32).

Parameter names match torch.nn.MultiheadAttention (in_proj_weight, in_proj_bias, out_proj.*) so
reference checkpoints load unchanged (vlf/lvf: models/vla.py:11; VOC: models/voc.py:66,123;
decoder self-attention: models/deformable_transformer.py:308).  Sequence-first layout [L,B,E].
The input/output projections are plain fp32 GEMMs (hipBLASLt through torch).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import hot_ops


class HipMultiheadAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.training:
            raise RuntimeError("HipMultiheadAttention is inference-only (no backward kernel)")
        E = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        if query is key:
            qk = F.linear(query, w[:2 * E], b[:2 * E])
            q, k = qk[..., :E], qk[..., E:]
            q, k = q.contiguous(), k.contiguous()
        else:
            q = F.linear(query, w[:E], b[:E])
            k = F.linear(key, w[E:2 * E], b[E:2 * E])
        v = F.linear(value, w[2 * E:], b[2 * E:])
        o = hot_ops.mha_core(q, k, v, self.num_heads, key_padding_mask)
        return self.out_proj(o)
