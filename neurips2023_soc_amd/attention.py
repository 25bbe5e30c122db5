"""Multi-head attention module whose core runs in the HIP kernel K3 (soc_xattn_f32).

Parameter names match torch.nn.MultiheadAttention (in_proj_weight, in_proj_bias, out_proj.*) so
reference checkpoints load unchanged (vlf/lvf: models/vla.py:11; VOC: models/voc.py:66,123;
decoder self-attention: models/deformable_transformer.py:308).  Sequence-first layout [L,B,E].
The input/output projections go through fused.linear: library GEMM for pixel-sized inputs, K7 for
query / word-sized ones, where the positional add in front (`query_add`, `key_add`) is folded in.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import fused, hot_ops


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
                key_padding_mask: Optional[torch.Tensor] = None, query_add: Optional[torch.Tensor] = None,
                key_add: Optional[torch.Tensor] = None, batch_first: bool = False,
                attn_mask: Optional[torch.Tensor] = None, post_norm: Optional[nn.LayerNorm] = None,
                out_mul: Optional[torch.Tensor] = None) -> torch.Tensor:
        """attention(query + query_add, key + key_add, value): the *_add terms are the positional
        embeddings the reference adds before calling nn.MultiheadAttention (with_pos_embed).
        batch_first: tensors are [B,L,E] instead of nn.MultiheadAttention's [L,B,E].
        attn_mask: additive float mask as in nn.MultiheadAttention ([Lq,Lk], [B,Lq,Lk] or [B*heads,Lq,Lk]).
        post_norm: return post_norm(query + attention(...)) -- the post-norm residual every caller on the query chain
        applies next -- with out_proj, the add and the LayerNorm in one launch (K16) when the rows are few.
        out_mul: return out_mul * attention(...) (reference models/vla.py:24 `tgt * tgt2`), the product fused into the
        output projection where that runs on K20."""
        if self.training:
            raise RuntimeError("HipMultiheadAttention is inference-only (no backward kernel)")
        E = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        wq, wk, wv = (w[:E], b[:E]), (w[E:2 * E], b[E:2 * E]), (w[2 * E:], b[2 * E:])
        if query is key and query_add is key_add:
            if fused.is_small(query):
                if value is query:       # self-attention: q, k, v in one launch (v without the pos add)
                    q, k, v = fused.linear_multi(query, [(*wq, True), (*wk, True), (*wv, False)], query_add)
                else:
                    q, k = fused.linear_multi(query, [(*wq, True), (*wk, True)], query_add)
                    v = fused.linear(value, *wv)
            else:
                x = query if query_add is None else query + query_add
                qk = F.linear(x, w[:2 * E], b[:2 * E])
                q, k = qk[..., :E].contiguous(), qk[..., E:].contiguous()
                v = fused.linear(value, *wv)
        else:
            q = fused.linear(query, *wq, add=query_add)
            if value is key and fused.is_small(key):
                k, v = fused.linear_multi(key, [(*wk, True), (*wv, False)], key_add)
            else:
                k = fused.linear(key, *wk, add=key_add)
                v = fused.linear(value, *wv)
        if attn_mask is None:    # (keeps the call shape the CPU plumbing tests patch in)
            o = hot_ops.mha_core(q, k, v, self.num_heads, key_padding_mask, batch_first=batch_first)
        else:
            o = hot_ops.mha_core(q, k, v, self.num_heads, key_padding_mask, batch_first=batch_first,
                                 attn_mask=attn_mask)
        if post_norm is None:
            return fused.linear(o, self.out_proj.weight, self.out_proj.bias, mul=out_mul)
        assert out_mul is None
        if o.shape == query.shape and hot_ops.row_mlp_supported(o, [self.out_proj.weight], has_ln=True):
            return hot_ops.row_mlp(o, [(self.out_proj.weight, self.out_proj.bias)], residual=query,
                                   ln=(post_norm.weight, post_norm.bias, post_norm.eps))
        return hot_ops.add_layernorm(query, fused.apply(self.out_proj, o), post_norm.weight, post_norm.bias,
                                     post_norm.eps, return_sum=False)[1]
