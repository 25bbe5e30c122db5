"""Video object cluster module (reference models/voc.py:181-335), inference form.

Encoder: self-attention + FFN over all T*Q frame queries (window_size 0 = full attention);
decoder: cross -> self -> FFN with queries initialised from the sentence feature.  Every
attention core is the HIP kernel K3; post-norm throughout (pre_norm=False in every config).
"""
from __future__ import annotations

import torch
from torch import nn

from . import hot_ops
from . import fused
from .fused import linear_relu
from .attention import HipMultiheadAttention


def _add_norm(x, y, norm: nn.LayerNorm):
    return hot_ops.add_layernorm(x, y, norm.weight, norm.bias, norm.eps, return_sum=False)[1]


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt):
        return _add_norm(tgt, fused.apply(self.linear2, linear_relu(tgt, self.linear1)), self.norm)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.self_attn = HipMultiheadAttention(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, tgt_key_padding_mask=None, query_pos=None):
        return _add_norm(tgt, self.self_attn(tgt, tgt, tgt, tgt_key_padding_mask, query_add=query_pos,
                                             key_add=query_pos), self.norm)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.multihead_attn = HipMultiheadAttention(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, memory, memory_key_padding_mask=None, pos=None, query_pos=None):
        return _add_norm(tgt, self.multihead_attn(tgt, memory, memory, memory_key_padding_mask,
                                                  query_add=query_pos, key_add=pos), self.norm)


class VOC(nn.Module):
    def __init__(self, config, pre_norm: bool = False, aux_loss: bool = False):
        super().__init__()
        if pre_norm:
            raise NotImplementedError("SOC always builds VOC with post-norm")
        if config["window_size"] != 0:
            raise NotImplementedError("temporal-window VOC (reference voc.py:356-414) is unused by every "
                                      "shipped config; SURVEY.md 8f rank 4")
        d = config["input_dim"]
        self.window_size = config["window_size"]
        self.num_frame_queries, self.num_queries = config["num_frame_queries"], config["num_queries"]
        self.num_heads, self.num_layers = config["nheads"], config["dec_layers"]
        self.num_frames, self.enc_layers = config["num_frames"], config["enc_layers"]
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        self.src_embed = nn.Identity()
        self.fq_pos = nn.Embedding(self.num_frame_queries, d)
        self.query_embed = nn.Embedding(self.num_queries, d)
        self.decoder_norm = nn.LayerNorm(d)
        if self.enc_layers > 0:
            self.enc_self_attn = nn.ModuleList(SelfAttentionLayer(d, self.num_heads) for _ in range(self.enc_layers))
            self.enc_ffn = nn.ModuleList(FFNLayer(d, config["dim_feedforward"]) for _ in range(self.enc_layers))
        for _ in range(self.num_layers):
            self.transformer_self_attention_layers.append(SelfAttentionLayer(d, self.num_heads))
            self.transformer_cross_attention_layers.append(CrossAttentionLayer(d, self.num_heads))
            self.transformer_ffn_layers.append(FFNLayer(d, config["dim_feedforward"]))

    def forward(self, frame_query: torch.Tensor, language_query: torch.Tensor) -> torch.Tensor:
        """frame_query [L,T,B,Q,C] (all decoder levels), language_query [B,C] -> [1,B,Q,C].

        Eval semantics of the reference (:274-275): only the LAST decoder level is clustered."""
        if self.training:
            raise RuntimeError("inference-only module")
        fq = frame_query[-1]                      # [T,B,Q,C]
        T, B, Q, C = fq.shape
        x = fq.permute(0, 2, 1, 3).reshape(T * Q, B, C)  # (t q) b c
        for attn, ffn in zip(self.enc_self_attn, self.enc_ffn):
            x = ffn(attn(x))
        dec_pos = self.fq_pos.weight[None, :, None, :].expand(T, -1, B, -1).reshape(T * Q, B, C)
        qe = self.query_embed.weight[:, None, :].expand(-1, B, -1)
        out = language_query[None].expand(self.num_queries, -1, -1)
        for cross, self_attn, ffn in zip(self.transformer_cross_attention_layers,
                                         self.transformer_self_attention_layers, self.transformer_ffn_layers):
            out = cross(out, x, pos=dec_pos, query_pos=qe)
            out = self_attn(out, query_pos=qe)
            out = ffn(out)
        return self.decoder_norm(out).transpose(0, 1)[None]
