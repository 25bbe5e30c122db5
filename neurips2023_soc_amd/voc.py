"""Video object cluster module (reference models/voc.py:181-414), inference form.

Encoder: self-attention + FFN over all T*Q frame queries (window_size 0 = full attention, every
shipped config) or over temporal windows of `window_size` frames (plain windows on even layers,
windows shifted by ceil(W/2) frames on odd ones, :336-414); decoder: cross -> self -> FFN with queries
initialised from the sentence feature.  Every attention core is the HIP kernel K3; post-norm
throughout (pre_norm=False in every config).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
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

    def forward(self, tgt, tgt_key_padding_mask=None, query_pos=None, tgt_mask=None):
        return self.self_attn(tgt, tgt, tgt, tgt_key_padding_mask, query_add=query_pos, key_add=query_pos,
                              attn_mask=tgt_mask, post_norm=self.norm)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.multihead_attn = HipMultiheadAttention(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, memory, memory_key_padding_mask=None, pos=None, query_pos=None):
        return self.multihead_attn(tgt, memory, memory, memory_key_padding_mask, query_add=query_pos, key_add=pos,
                                   post_norm=self.norm)


class VOC(nn.Module):
    def __init__(self, config, pre_norm: bool = False, aux_loss: bool = False):
        super().__init__()
        if pre_norm:
            raise NotImplementedError("SOC always builds VOC with post-norm")
        if config["window_size"] < 0:
            raise ValueError("window_size must be >= 0")
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

    def forward(self, frame_query: torch.Tensor, language_query: torch.Tensor, independent_clips: bool = False) -> torch.Tensor:
        """frame_query [L,T,B,Q,C] (all decoder levels), language_query [B,C] -> [1,B,Q,C].

        Eval semantics of the reference (:274-275): only the LAST decoder level is clustered.
        independent_clips: the B clips are separate inference requests that share a launch (graph_runner group pipelines): every
        clip is clustered over ITS OWN frames, i.e. exactly what B forwards with B = 1 compute, in one batched pass (the
        attention layers treat b as the batch dimension).  Default (False): the reference's B > 1 behaviour, see below."""
        if self.training:
            raise RuntimeError("inference-only module")
        fq = frame_query[-1]                      # [T,B,Q,C]
        T, B, Q, C = fq.shape
        # The reference RESHAPES [L,T,B,Q,C] to [L*B,T,Q,C] (:279) instead of permuting: for B > 1 the (t, b)
        # pairs are re-read as (b, t) in memory order, i.e. frames are redistributed over the batch.
        # Identity for B = 1 (every inference driver); kept so that a padded batch matches the reference too.
        if not independent_clips:
            fq = fq.reshape(B, T, Q, C).transpose(0, 1)
        if self.enc_layers == 0:
            x = fq.permute(0, 2, 1, 3).reshape(T * Q, B, C)
        elif self.window_size == 0:
            x = fq.permute(0, 2, 1, 3).reshape(T * Q, B, C)  # (t q) b c
            for attn, ffn in zip(self.enc_self_attn, self.enc_ffn):
                x = ffn(attn(x))
        else:
            x = self._encode_windows(fq.permute(0, 2, 1, 3))
        dec_pos = self.fq_pos.weight[None, :, None, :].expand(T, -1, B, -1).reshape(T * Q, B, C)
        qe = self.query_embed.weight[:, None, :].expand(-1, B, -1)
        out = language_query[None].expand(self.num_queries, -1, -1)
        for cross, self_attn, ffn in zip(self.transformer_cross_attention_layers,
                                         self.transformer_self_attention_layers, self.transformer_ffn_layers):
            out = cross(out, x, pos=dec_pos, query_pos=qe)
            out = self_attn(out, query_pos=qe)
            out = ffn(out)
        return self.decoder_norm(out).transpose(0, 1)[None]

    # ------------------------------------------------------------------ temporal windows (:336-414)
    @staticmethod
    def window_masks(pad: torch.Tensor, W: int, Q: int):
        """pad [B,T_] bool (True = frame added to fill the last window) -> (key padding mask of the plain
        windows [B*Nw, W*Q], additive mask of the shifted windows [B*Nw, W*Q, W*Q]: -1000 between frames that
        only became neighbours through the cyclic shift, or that are padding)."""
        B, T_ = pad.shape
        Nw, half = T_ // W, math.ceil(W / 2)
        plain = pad.view(B * Nw, W, 1).expand(-1, -1, Q).reshape(B * Nw, W * Q)
        by_query = torch.roll(pad, half, 1).view(B, Nw, W, 1).expand(-1, -1, -1, W)      # blocked rows
        blocked = by_query.clone()
        for n in (0, Nw - 1):                                   # first / last window: rows and columns
            blocked[:, n] = blocked[:, n] | blocked[:, n].transpose(-2, -1)
        blocked[:, 0, :half, half:] = True                       # window 0 mixes the clip's end with its start
        blocked[:, 0, half:, :half] = True
        frames = blocked.view(B * Nw, W, 1, W, 1).expand(-1, -1, Q, -1, Q).reshape(B * Nw, W * Q, W * Q)
        return plain, frames.to(torch.float32) * -1000.0

    def _encode_windows(self, x: torch.Tensor) -> torch.Tensor:
        """x [T,Q,B,C] -> (t q) b c after the windowed encoder (eval: even layers plain, odd layers shifted)."""
        T, Q, B, C = x.shape
        W = self.window_size
        T_ = math.ceil(T / W) * W
        Nw, half = T_ // W, math.ceil(W / 2)
        x = F.pad(x, (0, 0, 0, 0, 0, 0, 0, T_ - T))
        pad = torch.ones(B, T_, dtype=torch.bool, device=x.device)
        pad[:, :T] = False
        plain, shifted = self.window_masks(pad, W, Q)

        def split(t):     # [T_,Q,B,C] -> [(W Q), (B Nw), C]: every window is a batch element
            return t.view(Nw, W, Q, B, C).permute(1, 2, 3, 0, 4).reshape(W * Q, B * Nw, C)

        def merge(t):
            return t.reshape(W, Q, B, Nw, C).permute(3, 0, 1, 2, 4).reshape(T_, Q, B, C)

        for i, (attn, ffn) in enumerate(zip(self.enc_self_attn, self.enc_ffn)):
            if i % 2 == 0:
                x = merge(ffn(attn(split(x), tgt_key_padding_mask=plain)))
            else:
                x = torch.roll(merge(ffn(attn(split(torch.roll(x, half, 0)), tgt_mask=shifted))), -half, 0)
        return x[:T].reshape(T * Q, B, C)
