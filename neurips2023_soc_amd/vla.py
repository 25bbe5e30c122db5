"""Video-language alignment block (reference models/vla.py:8-24): tgt * MHA(tgt, memory+pos, memory)."""
from __future__ import annotations

from typing import Optional

from torch import Tensor, nn

from .attention import HipMultiheadAttention


class MMF(nn.Module):
    def __init__(self, d_model: int, nhead: int, dropout: float = 0.0):
        super().__init__()
        self.multihead_attn = HipMultiheadAttention(d_model, nhead, dropout)

    def forward(self, tgt: Tensor, memory: Tensor, memory_key_padding_mask: Optional[Tensor] = None,
                pos: Optional[Tensor] = None, query_pos: Optional[Tensor] = None, batch_first: bool = False) -> Tensor:
        # batch_first: tgt [B,Lq,C], memory / pos [B,Lk,C] (a launch group of clips: the tokens stay in their '(b t) (h w) c' storage)
        # `tgt *` is applied in the epilogue of the output projection where that is a K20 launch (pixel-sized tgt)
        return self.multihead_attn(tgt, memory, memory, memory_key_padding_mask, query_add=query_pos, key_add=pos,
                                   out_mul=tgt, batch_first=batch_first)
