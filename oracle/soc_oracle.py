"""CPU oracle for SOC's per-clip eval forward -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (neurips2023_soc_amd/) never does and has no CPU fallback.

It is a functional restatement (plain torch-CPU fp32 ops over a ``state_dict``) of what the
reference computes on the path SURVEY.md section 8 names; every function cites the reference
file:line it follows.  Pinning: the reference has no golden vectors for this path,
so the oracle is pinned against outputs of the reference itself, run in the build container by
tests/golden/make_goldens.py and committed under tests/golden/*.npz
(tests/test_oracle_vs_golden.py checks every one of them).

The four hot ops are restated in the *kernel-boundary* form the HIP library exposes
(include/soc_hip.h), so the same functions serve as per-kernel checkers:
  msda_core            <- models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84,237-299
  window_attention_core<- models/video_swin_transformer.py:138-169,215-249,316-329
  mha_core             <- torch.nn.MultiheadAttention as called at models/vla.py:16-24 etc.
  dynamic_mask_core    <- models/soc.py:399-483,536-549
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

SWIN_CFGS = {  # reference models/video_swin_transformer.py:733-779
    "video-swin-t": dict(embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24)),
    "video-swin-s": dict(embed_dim=96, depths=(2, 2, 18, 2), heads=(3, 6, 12, 24)),
    "video-swin-b": dict(embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32)),
}
WINDOW = (8, 7, 7)


# ----------------------------------------------------------------------------- small helpers
def linear(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def layer_norm(sd: SD, p: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def group_norm(sd: SD, p: str, x: Tensor, groups: int) -> Tensor:
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def inverse_sigmoid(x: Tensor, eps: float = 1e-5) -> Tensor:
    """reference misc.py:427-431"""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def mlp_relu(sd: SD, p: str, x: Tensor, n: int) -> Tensor:
    """reference models/soc.py:552-564"""
    for i in range(n):
        x = linear(sd, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def resize_mask(mask: Tensor, size: Tuple[int, int]) -> Tensor:
    """nearest resize of a bool pad mask [N,H,W] (reference video_swin_transformer.py:712-714)"""
    return F.interpolate(mask[None].float(), size=size).to(torch.bool)[0]


def sine_pos_2d(mask: Tensor, num_pos_feats: int = 128, temperature: float = 10000.0) -> Tensor:
    """reference models/position_encoding.py:46-82 (normalize=True) -> [N, 2*num_pos_feats, H, W]"""
    not_mask = ~mask
    y = not_mask.cumsum(1, dtype=torch.float32)
    x = not_mask.cumsum(2, dtype=torch.float32)
    scale, eps = 2 * math.pi, 1e-6
    y = (y - 0.5) / (y[:, -1:, :] + eps) * scale
    x = (x - 0.5) / (x[:, :, -1:] + eps) * scale
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    d = temperature ** (2 * (d // 2) / num_pos_feats)
    px = x[..., None] / d
    py = y[..., None] / d
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def sine_pos_1d(mask: Tensor, num_pos_feats: int = 256, temperature: float = 10000.0) -> Tensor:
    """reference models/position_encoding.py:11-44 (normalize=True): mask [B,L] -> [B,C,L]"""
    x = (~mask).cumsum(1, dtype=torch.float32)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    d = temperature ** (2 * (d // 2) / num_pos_feats)
    p = x[:, :, None] / d
    p = torch.stack((p[:, :, 0::2].sin(), p[:, :, 1::2].cos()), dim=3).flatten(2)
    return p.permute(0, 2, 1)


# ----------------------------------------------------------------------------- hot op 3: MHA core
def mha_core(q: Tensor, k: Tensor, v: Tensor, n_heads: int,
             key_padding_mask: Optional[Tensor] = None, batch_first: bool = False,
             attn_mask: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T / sqrt(d)) v per head, seq-first layout ([B,L,E] tensors when batch_first).

    q [Lq,B,E], k/v [Lk,B,E] are the *projected* tensors; key_padding_mask [B,Lk] bool, True =
    ignore.  Follows torch.nn.functional.multi_head_attention_forward (q is pre-scaled by
    1/sqrt(d), padded keys get -inf) as invoked by reference models/vla.py:20-23,
    models/voc.py:89-90,146-149 and models/deformable_transformer.py:333.
    """
    if batch_first:
        return mha_core(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), n_heads,
                        key_padding_mask, attn_mask=attn_mask).transpose(0, 1)
    Lq, B, E = q.shape
    Lk = k.shape[0]
    hd = E // n_heads
    qh = (q * math.sqrt(1.0 / hd)).reshape(Lq, B * n_heads, hd).transpose(0, 1)
    kh = k.reshape(Lk, B * n_heads, hd).transpose(0, 1)
    vh = v.reshape(Lk, B * n_heads, hd).transpose(0, 1)
    s = torch.bmm(qh, kh.transpose(1, 2))
    if attn_mask is not None:  # additive float mask [Lq,Lk] / [B,Lq,Lk] / [B*nH,Lq,Lk] (F.multi_head_attention_forward)
        am = attn_mask if attn_mask.dim() == 3 else attn_mask[None].expand(B, -1, -1)
        if am.shape[0] == B and n_heads > 1:
            am = am[:, None].expand(B, n_heads, Lq, Lk).reshape(B * n_heads, Lq, Lk)
        s = s + am
    if key_padding_mask is not None:
        m = key_padding_mask.view(B, 1, 1, Lk).expand(B, n_heads, 1, Lk).reshape(B * n_heads, 1, Lk)
        s = s.masked_fill(m, float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = torch.bmm(a, vh)  # [B*nH, Lq, hd]
    return o.transpose(0, 1).reshape(Lq, B, E)


def mha(sd: SD, p: str, query: Tensor, key: Tensor, value: Tensor, n_heads: int = 8,
        key_padding_mask: Optional[Tensor] = None, attn_mask: Optional[Tensor] = None) -> Tensor:
    """nn.MultiheadAttention (in_proj -> core -> out_proj), parameters under prefix ``p``."""
    E = query.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(query, w[:E], b[:E])
    k = F.linear(key, w[E:2 * E], b[E:2 * E])
    v = F.linear(value, w[2 * E:], b[2 * E:])
    o = mha_core(q, k, v, n_heads, key_padding_mask, attn_mask=attn_mask)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def mmf(sd: SD, p: str, tgt: Tensor, memory: Tensor, memory_key_padding_mask: Optional[Tensor],
        pos: Optional[Tensor]) -> Tensor:
    """VLA fusion: tgt * MHA(tgt, memory+pos, memory)  (reference models/vla.py:16-24)"""
    key = memory if pos is None else memory + pos
    return tgt * mha(sd, p + ".multihead_attn", tgt, key, memory, 8, memory_key_padding_mask)


# ----------------------------------------------------------------------------- hot op 1: 3-D window attention
def clamp_window(size: Sequence[int], window: Sequence[int], shift: Sequence[int]):
    """reference models/video_swin_transformer.py:71-84"""
    w, s = list(window), list(shift)
    for i in range(3):
        if size[i] <= window[i]:
            w[i] = size[i]
            s[i] = 0
    return tuple(w), tuple(s)


def relative_position_index(window: Sequence[int] = WINDOW) -> Tensor:
    """[N,N] int64 index into the (2Wd-1)(2Wh-1)(2Ww-1) bias table (reference :113-128)"""
    wd, wh, ww = window
    zz, yy, xx = torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")
    c = torch.stack([zz.flatten(), yy.flatten(), xx.flatten()])  # [3,N]
    rel = c[:, :, None] - c[:, None, :]
    return ((rel[0] + wd - 1) * ((2 * wh - 1) * (2 * ww - 1)) + (rel[1] + wh - 1) * (2 * ww - 1)
            + (rel[2] + ww - 1))


def shift_region_ids(dims: Sequence[int], window: Sequence[int], shift: Sequence[int]) -> Tensor:
    """Region id per position of the shifted, padded volume [Dp,Hp,Wp] (reference :316-324).

    Later slice assignments overwrite earlier ones exactly as the reference's nested loops do
    (this matters when a shift is 0: slice(-0, None) covers the whole axis).
    """
    ids = torch.zeros(tuple(dims), dtype=torch.float32)
    cnt = 0
    sl = [(slice(-window[i]), slice(-window[i], -shift[i]), slice(-shift[i], None)) for i in range(3)]
    for d in sl[0]:
        for h in sl[1]:
            for w in sl[2]:
                ids[d, h, w] = cnt
                cnt += 1
    return ids


def window_attention_core(qkv: Tensor, qkv_bias: Tensor, bias_table: Tensor, n_heads: int,
                          window: Sequence[int] = WINDOW, shift: Sequence[int] = (0, 0, 0)) -> Tensor:
    """Kernel-boundary form of (shifted) 3-D window attention.

    qkv [B,D,H,W,3C] = qkv Linear applied to the *unpadded* normed tokens; qkv_bias [3C] is what
    a zero (padded) token projects to.  Returns the attention output before ``proj`` in token
    layout [B,D,H,W,C].  Equivalent to reference SwinTransformerBlock3D.forward_part1 :219-249
    between ``norm1``/``qkv`` and ``proj`` because qkv/proj are per-token linears:
    pad -> roll(-shift) -> partition -> softmax(q*scale k^T + B[idx[:N,:N]] + mask) v ->
    reverse -> roll(+shift) -> crop.  ``window``/``shift`` are the *requested* values; clamping
    (:71-84) happens here.
    """
    B, D, H, W, C3 = qkv.shape
    C = C3 // 3
    hd = C // n_heads
    win, sh = clamp_window((D, H, W), window, shift)
    pd = (win[0] - D % win[0]) % win[0]
    pb = (win[1] - H % win[1]) % win[1]
    pr = (win[2] - W % win[2]) % win[2]
    Dp, Hp, Wp = D + pd, H + pb, W + pr
    x = qkv.new_empty(B, Dp, Hp, Wp, C3)
    x[:] = qkv_bias  # zero tokens -> bias (reference pads AFTER norm1, :219-225)
    x[:, :D, :H, :W] = qkv
    shifted = any(s > 0 for s in sh)
    if shifted:
        x = torch.roll(x, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
    nd, nh, nw = Dp // win[0], Hp // win[1], Wp // win[2]
    N = win[0] * win[1] * win[2]
    xw = x.view(B, nd, win[0], nh, win[1], nw, win[2], C3).permute(0, 1, 3, 5, 2, 4, 6, 7)
    xw = xw.reshape(B * nd * nh * nw, N, 3, n_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * (hd ** -0.5), xw[1], xw[2]
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(window)[:N, :N].reshape(-1)
    attn = attn + bias_table[idx].view(N, N, n_heads).permute(2, 0, 1)[None]
    if shifted:
        ids = shift_region_ids((Dp, Hp, Wp), win, sh)
        ids = ids.view(nd, win[0], nh, win[1], nw, win[2]).permute(0, 2, 4, 1, 3, 5).reshape(-1, N)
        diff = ids[:, None, :] - ids[:, :, None]
        m = torch.zeros_like(diff).masked_fill(diff != 0, -100.0)  # -100, not -inf (:328)
        nW = m.shape[0]
        attn = (attn.view(B, nW, n_heads, N, N) + m[None, :, None]).view(-1, n_heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, nd, nh, nw, win[0], win[1], win[2], C)
    o = o.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, Dp, Hp, Wp, C)
    if shifted:
        o = torch.roll(o, shifts=sh, dims=(1, 2, 3))
    return o[:, :D, :H, :W].contiguous()


def swin_block(sd: SD, p: str, x: Tensor, n_heads: int, shift: Sequence[int]) -> Tensor:
    """reference SwinTransformerBlock3D.forward :254-274 (eval: DropPath = identity)"""
    h = layer_norm(sd, p + ".norm1", x)
    qkv = linear(sd, p + ".attn.qkv", h)
    a = window_attention_core(qkv, sd[p + ".attn.qkv.bias"], sd[p + ".attn.relative_position_bias_table"],
                              n_heads, WINDOW, shift)
    x = x + linear(sd, p + ".attn.proj", a)
    h = layer_norm(sd, p + ".norm2", x)
    h = linear(sd, p + ".mlp.fc2", F.gelu(linear(sd, p + ".mlp.fc1", h)))
    return x + h


def patch_merging(sd: SD, p: str, x: Tensor) -> Tensor:
    """reference PatchMerging.forward :290-312, x [B,D,H,W,C]"""
    H, W = x.shape[2], x.shape[3]
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    return F.linear(layer_norm(sd, p + ".norm", x), sd[p + ".reduction.weight"])


def swin_backbone(sd: SD, clip: Tensor, backbone: str = "video-swin-t",
                  p: str = "backbone.0.body") -> List[Tensor]:
    """clip [T,3,H,W] (B=1) -> 4 per-frame pyramids [(T,C_l,H_l,W_l)]

    reference VideoSwinTransformerBackbone.forward :678-697, PatchEmbed3D :438-456,
    BasicLayer.forward :392-413.
    """
    cfg = SWIN_CFGS[backbone]
    x = clip.permute(1, 0, 2, 3)[None]  # [1,3,T,H,W]
    H, W = x.shape[-2:]
    if W % 4:
        x = F.pad(x, (0, 4 - W % 4))
    if H % 4:
        x = F.pad(x, (0, 0, 0, 4 - H % 4))
    x = F.conv3d(x, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=(1, 4, 4))
    x = x.permute(0, 2, 3, 4, 1)  # [1,T,h,w,C]
    x = layer_norm(sd, p + ".patch_embed.norm", x)
    outs = []
    shift = tuple(w // 2 for w in WINDOW)
    for li, (depth, heads) in enumerate(zip(cfg["depths"], cfg["heads"])):
        for bi in range(depth):
            x = swin_block(sd, f"{p}.layers.{li}.blocks.{bi}", x, heads, (0, 0, 0) if bi % 2 == 0 else shift)
        outs.append(x[0].permute(0, 3, 1, 2).contiguous())  # [T,C,h,w]
        if li < 3:
            x = patch_merging(sd, f"{p}.downsamples.{li}", x)
    return outs


# ----------------------------------------------------------------------------- hot op 2: MSDA core
def msda_core(value: Tensor, shapes: Tensor, level_start: Tensor, loc: Tensor, w: Tensor) -> Tensor:
    """Multi-scale deformable attention sampling core -> [N, Lq, M*D].

    value [N,S,M,D], shapes [L,2] (H,W), level_start [L], loc [N,Lq,M,L,P,2] (x,y in [0,1]),
    w [N,Lq,M,L,P].  Follows the CUDA kernel's rules literally
    (reference ms_deform_im2col_cuda.cuh:33-84,237-299): pixel coords h=y*H-0.5, w=x*W-0.5; a
    point contributes only if -1<h<H and -1<w<W; each of the 4 taps is zero when out of range;
    tap weights (1-lh)(1-lw) etc.  Accumulation order: levels, then points.
    """
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = value.new_zeros(N, Lq, M, D)
    n_idx = torch.arange(N).view(N, 1, 1, 1).expand(N, Lq, M, P)
    m_idx = torch.arange(M).view(1, 1, M, 1).expand(N, Lq, M, P)
    for l in range(L):
        Hl, Wl = int(shapes[l, 0]), int(shapes[l, 1])
        base = int(level_start[l])
        hh = loc[:, :, :, l, :, 1] * Hl - 0.5
        ww = loc[:, :, :, l, :, 0] * Wl - 0.5
        ok = (hh > -1) & (ww > -1) & (hh < Hl) & (ww < Wl)
        h0 = torch.floor(hh)
        w0 = torch.floor(ww)
        lh, lw = hh - h0, ww - w0
        h0, w0 = h0.long(), w0.long()
        acc = value.new_zeros(N, Lq, M, P, D)
        for dh, dw, tw in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                           (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hi, wi = h0 + dh, w0 + dw
            inb = ok & (hi >= 0) & (hi <= Hl - 1) & (wi >= 0) & (wi <= Wl - 1)
            flat = base + hi.clamp(0, Hl - 1) * Wl + wi.clamp(0, Wl - 1)
            tap = value[n_idx, flat, m_idx]  # [N,Lq,M,P,D]
            acc = acc + tap * (tw * inb)[..., None]
        out = out + (acc * w[:, :, :, l, :, None]).sum(3)
    return out.view(N, Lq, M * D)


def msda_fused_core(value: Tensor, shapes: Tensor, level_start: Tensor, ref: Tensor, offsets: Tensor,
                    logits: Tensor, pad_mask: Optional[Tensor] = None) -> Tensor:
    """Kernel-boundary form of the fused K2 entry point: what MSDeformAttn.forward does between
    value_proj and output_proj (reference models/ops/modules/ms_deform_attn.py:95-114)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = offsets.shape
    if pad_mask is not None:
        value = value.masked_fill(pad_mask.bool()[:, :, None, None], 0.0)
    aw = torch.softmax(logits.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    if ref.shape[-1] == 2:
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(value.dtype)
        loc = ref[:, :, None, :, None, :] + offsets / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + offsets / P * ref[:, :, None, :, None, 2:] * 0.5
    return msda_core(value.contiguous(), shapes, level_start, loc.contiguous(), aw.contiguous())


def msda_module(sd: SD, p: str, query: Tensor, ref: Tensor, src: Tensor, shapes: Tensor,
                level_start: Tensor, pad_mask: Optional[Tensor]) -> Tensor:
    """reference MSDeformAttn.forward models/ops/modules/ms_deform_attn.py:79-117"""
    N, Lq, _ = query.shape
    S = src.shape[1]
    M, L, P = 8, shapes.shape[0], 4
    value = linear(sd, p + ".value_proj", src)
    if pad_mask is not None:
        value = value.masked_fill(pad_mask[..., None], 0.0)
    value = value.view(N, S, M, -1)
    off = linear(sd, p + ".sampling_offsets", query).view(N, Lq, M, L, P, 2)
    aw = torch.softmax(linear(sd, p + ".attention_weights", query).view(N, Lq, M, L * P), -1)
    aw = aw.view(N, Lq, M, L, P)
    if ref.shape[-1] == 2:
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(query.dtype)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    out = msda_core(value.contiguous(), shapes, level_start, loc.contiguous(), aw.contiguous())
    return linear(sd, p + ".output_proj", out)


# ----------------------------------------------------------------------------- deformable transformer
def valid_ratio(mask: Tensor) -> Tensor:
    """reference models/deformable_transformer.py:123-130, mask [N,H,W] -> [N,2] (w,h)"""
    _, H, W = mask.shape
    vh = (~mask[:, :, 0]).sum(1).float() / H
    vw = (~mask[:, 0, :]).sum(1).float() / W
    return torch.stack([vw, vh], -1)


def encoder_reference_points(shapes: Tensor, ratios: Tensor) -> Tensor:
    """reference :273-285 -> [N, S, L, 2]"""
    pts = []
    for l in range(shapes.shape[0]):
        Hl, Wl = int(shapes[l, 0]), int(shapes[l, 1])
        ry, rx = torch.meshgrid(torch.linspace(0.5, Hl - 0.5, Hl), torch.linspace(0.5, Wl - 0.5, Wl),
                                indexing="ij")
        ry = ry.reshape(-1)[None] / (ratios[:, None, l, 1] * Hl)
        rx = rx.reshape(-1)[None] / (ratios[:, None, l, 0] * Wl)
        pts.append(torch.stack((rx, ry), -1))
    return torch.cat(pts, 1)[:, :, None] * ratios[:, None]


def deformable_transformer(sd: SD, srcs: List[Tensor], masks: List[Tensor], poses: List[Tensor],
                           query_embed: Tensor, bbox_prefix: str = "bbox_embed",
                           p: str = "transformer", n_enc: int = 3, n_dec: int = 3):
    """reference DeformableTransformer.forward :132-220 (two_stage=False, with_box_refine=True).

    Returns hs [n_dec,N,Q,C], memory maps (3 finest levels), init_ref [N,Q,2], inter_refs [n_dec,N,Q,4].
    The decoder's top-30 sample bookkeeping (:383-389) is unused by SOC.forward and omitted.
    """
    src_f, mask_f, pos_f, shp = [], [], [], []
    for l, (s, m, pe) in enumerate(zip(srcs, masks, poses)):
        shp.append(s.shape[-2:])
        src_f.append(s.flatten(2).transpose(1, 2))
        mask_f.append(m.flatten(1))
        pos_f.append(pe.flatten(2).transpose(1, 2) + sd[p + ".level_embed"][l].view(1, 1, -1))
    src, mask, pos = torch.cat(src_f, 1), torch.cat(mask_f, 1), torch.cat(pos_f, 1)
    shapes = torch.as_tensor([list(s) for s in shp], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    ratios = torch.stack([valid_ratio(m) for m in masks], 1)

    ref_enc = encoder_reference_points(shapes, ratios)
    x = src
    for i in range(n_enc):  # encoder layer :253-263
        q = f"{p}.encoder.layers.{i}"
        x = layer_norm(sd, q + ".norm1", x + msda_module(sd, q + ".self_attn", x + pos, ref_enc, x, shapes, lsi, mask))
        x = layer_norm(sd, q + ".norm2", x + linear(sd, q + ".linear2", F.relu(linear(sd, q + ".linear1", x))))
    memory = x

    N = memory.shape[0]
    qpos = query_embed[None].expand(N, -1, -1)
    ref = linear(sd, p + ".reference_points", qpos).sigmoid()
    init_ref = ref
    tgt = torch.zeros_like(qpos)
    hs, refs = [], []
    for i in range(n_dec):  # decoder :361-412, layer :330-347
        q = f"{p}.decoder.layers.{i}"
        if ref.shape[-1] == 4:
            ref_in = ref[:, :, None] * torch.cat([ratios, ratios], -1)[:, None]
        else:
            ref_in = ref[:, :, None] * ratios[:, None]
        qk = (tgt + qpos).transpose(0, 1)
        t2 = mha(sd, q + ".self_attn", qk, qk, tgt.transpose(0, 1), 8).transpose(0, 1)
        tgt = layer_norm(sd, q + ".norm2", tgt + t2)
        t2 = msda_module(sd, q + ".cross_attn", tgt + qpos, ref_in, memory, shapes, lsi, mask)
        tgt = layer_norm(sd, q + ".norm1", tgt + t2)
        tgt = layer_norm(sd, q + ".norm3", tgt + linear(sd, q + ".linear2", F.relu(linear(sd, q + ".linear1", tgt))))
        tmp = mlp_relu(sd, f"{bbox_prefix}.{i}", tgt, 3)  # shared with the heads (soc.py:91-95)
        if ref.shape[-1] == 4:
            new_ref = (tmp + inverse_sigmoid(ref)).sigmoid()
        else:
            tmp = tmp.clone()
            tmp[..., :2] = tmp[..., :2] + inverse_sigmoid(ref)
            new_ref = tmp.sigmoid()
        ref = new_ref
        hs.append(tgt)
        refs.append(ref)

    mem_maps, at = [], 0
    for l in range(len(srcs) - 1):
        Hl, Wl = shp[l]
        mem_maps.append(memory[:, at:at + Hl * Wl].reshape(N, Hl, Wl, -1).permute(0, 3, 1, 2).contiguous())
        at += Hl * Wl
    return torch.stack(hs), mem_maps, init_ref, torch.stack(refs)


# ----------------------------------------------------------------------------- VOC
def _voc_enc_layer(sd: SD, p: str, i: int, x: Tensor, key_padding_mask=None, attn_mask=None) -> Tensor:
    """SelfAttentionLayer.forward_post (:84-94) + FFNLayer.forward_post (:44-48) of encoder layer i"""
    a = f"{p}.enc_self_attn.{i}"
    x = layer_norm(sd, a + ".norm", x + mha(sd, a + ".self_attn", x, x, x, 8, key_padding_mask, attn_mask))
    f = f"{p}.enc_ffn.{i}"
    return layer_norm(sd, f + ".norm", x + linear(sd, f + ".linear2", F.relu(linear(sd, f + ".linear1", x))))


def voc_window_masks(pad: Tensor, W: int, fQ: int) -> Tuple[Tensor, Tensor]:
    """Masks of the temporal-window encoder (reference models/voc.py:361-377).

    pad [LB, T_] bool (True = padded frame).  Returns (key padding mask of the plain windows
    [LB*Nw, W*fQ] bool, additive mask of the shifted windows [LB*Nw, W*fQ, W*fQ] float: -1000 where a
    query frame may not see a key frame)."""
    LB, T_ = pad.shape
    Nw, half = T_ // W, int(math.ceil(W / 2))
    win = pad.view(LB * Nw, W)[..., None].repeat(1, 1, fQ).flatten(1)
    r = torch.roll(pad, half, 1).view(LB, Nw, W)[..., None].repeat(1, 1, 1, W)      # [LB,Nw,Wq,Wk], by QUERY frame
    r[:, 0] = r[:, 0] | r[:, 0].transpose(-2, -1)
    r[:, -1] = r[:, -1] | r[:, -1].transpose(-2, -1)
    r[:, 0, :half, half:] = True          # frames rolled in from the end of the clip ...
    r[:, 0, half:, :half] = True          # ... and the first real frames do not see each other
    tok = r.view(LB * Nw, W, 1, W, 1).repeat(1, 1, fQ, 1, fQ).view(LB * Nw, W * fQ, W * fQ)
    return win, tok.float() * -1000


def voc(sd: SD, hs_last: Tensor, sentence: Tensor, p: str = "voc", n_enc: int = 3, n_dec: int = 3,
        window_size: int = 0) -> Tensor:
    """reference VOC.forward models/voc.py:268-335 in eval (window_size = 0: full attention over all
    T*Q frame queries; > 0: temporal windows, plain on even encoder layers, shifted by ceil(W/2) frames on
    odd ones, :336-414).

    hs_last [T,B,Q,C] (= hs[-1]), sentence [B,C] -> [B,Q,C]
    """
    T, B, Q, C = hs_last.shape
    hs_last = hs_last.reshape(B, T, Q, C).transpose(0, 1)   # the reference reshapes [L,T,B,..] to [L*B,T,..] (:279); identity for B = 1
    if window_size == 0:
        fq = hs_last.permute(0, 2, 1, 3).reshape(T * Q, B, C)  # (t q) b c
        for i in range(n_enc):  # :349-351
            fq = _voc_enc_layer(sd, p, i, fq)
    else:
        W = window_size
        T_ = int(math.ceil(T / W)) * W
        x = F.pad(hs_last.permute(0, 2, 1, 3), (0, 0, 0, 0, 0, 0, 0, T_ - T))          # [T_,Q,B,C]
        pad = torch.ones(B, T_, dtype=torch.bool)
        pad[:, :T] = False
        win_mask, shift_mask = voc_window_masks(pad, W, Q)
        Nw, half = T_ // W, int(math.ceil(W / 2))

        def to_windows(t):   # [T_,Q,B,C] -> [(W Q), (B Nw), C]
            return t.view(Nw, W, Q, B, C).permute(1, 2, 3, 0, 4).reshape(W * Q, B * Nw, C)

        def from_windows(t):
            return t.reshape(W, Q, B, Nw, C).permute(3, 0, 1, 2, 4).reshape(T_, Q, B, C)

        for i in range(n_enc):
            if i % 2 == 0:
                x = from_windows(_voc_enc_layer(sd, p, i, to_windows(x), key_padding_mask=win_mask))
            else:
                y = _voc_enc_layer(sd, p, i, to_windows(torch.roll(x, half, 0)), attn_mask=shift_mask)
                x = torch.roll(from_windows(y), -half, 0)
        fq = x[:T].flatten(0, 1)
    dec_pos = sd[p + ".fq_pos.weight"][None, :, None, :].repeat(T, 1, B, 1).flatten(0, 1)
    qe = sd[p + ".query_embed.weight"][:, None, :].repeat(1, B, 1)
    nq = qe.shape[0]
    out = sentence[None].repeat(nq, 1, 1)  # :303-305
    for i in range(n_dec):  # cross -> self -> ffn :308-326
        c = f"{p}.transformer_cross_attention_layers.{i}"
        out = layer_norm(sd, c + ".norm", out + mha(sd, c + ".multihead_attn", out + qe, fq + dec_pos, fq, 8))
        s = f"{p}.transformer_self_attention_layers.{i}"
        out = layer_norm(sd, s + ".norm", out + mha(sd, s + ".self_attn", out + qe, out + qe, out, 8))
        f = f"{p}.transformer_ffn_layers.{i}"
        out = layer_norm(sd, f + ".norm", out + linear(sd, f + ".linear2", F.relu(linear(sd, f + ".linear1", out))))
    return layer_norm(sd, p + ".decoder_norm", out).transpose(0, 1)


# ----------------------------------------------------------------------------- FPN + mask head
def conv(sd: SD, p: str, x: Tensor, **kw) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], **kw)


def fpn_spatial_decoder(sd: SD, x: Tensor, feats: List[Tensor], p: str = "spatial_decoder") -> Tensor:
    """reference FPNSpatialDecoder.forward models/segmentation.py:46-74"""
    x = F.relu(group_norm(sd, p + ".gn1", conv(sd, p + ".lay1", x, padding=1), 8))
    x = F.relu(group_norm(sd, p + ".gn2", conv(sd, p + ".lay2", x, padding=1), 8))
    for i, (lay, gn, ad) in enumerate((("lay3", "gn3", "adapter1"), ("lay4", "gn4", "adapter2"),
                                        ("lay5", "gn5", "adapter3"))):
        cur = conv(sd, f"{p}.{ad}", feats[i])
        x = cur + F.interpolate(x, size=cur.shape[-2:], mode="nearest")
        x = F.relu(group_norm(sd, f"{p}.{gn}", conv(sd, f"{p}.{lay}", x, padding=1), 8))
    return conv(sd, p + ".out_lay", x, padding=1)


def dynamic_mask_core(feats: Tensor, params: Tensor, refs: Tensor, img_hw: Tuple[float, float],
                      stride: int = 4) -> Tensor:
    """Per-instance 3-layer dynamic 1x1 conv over [8 feature ch, rel_x, rel_y] -> [T*Q, h, w].

    feats [T,8,h,w]; params [T*Q,169] split as w0(8x10) w1(8x8) w2(1x8) b0(8) b1(8) b2(1), weights
    row-major [out][in]; refs [T*Q,2] normalised (x,y); instance order (t,q).  Relative coords are
    ref*(W_img,H_img) - (stride*x + stride//2, stride*y + stride//2)
    (reference models/soc.py:399-483,486-509,536-549; B=1).
    """
    T, Cm, h, w = feats.shape
    TQ = params.shape[0]
    Q = TQ // T
    ys = torch.arange(h, dtype=torch.float32) * stride + stride // 2
    xs = torch.arange(w, dtype=torch.float32) * stride + stride // 2
    rp = refs * torch.tensor([float(img_hw[1]), float(img_hw[0])])
    rx = rp[:, 0].view(TQ, 1, 1) - xs.view(1, 1, w)
    ry = rp[:, 1].view(TQ, 1, 1) - ys.view(1, h, 1)
    f = feats[:, None].expand(T, Q, Cm, h, w).reshape(TQ, Cm, h, w)
    x = torch.cat([f, rx.expand(TQ, h, w)[:, None], ry.expand(TQ, h, w)[:, None]], 1)  # [TQ,10,h,w]
    n0, n1 = (Cm + 2) * 8, 64
    w0 = params[:, :n0].view(TQ, 8, Cm + 2)
    w1 = params[:, n0:n0 + n1].view(TQ, 8, 8)
    w2 = params[:, n0 + n1:n0 + n1 + 8].view(TQ, 1, 8)
    o = n0 + n1 + 8
    b0, b1, b2 = params[:, o:o + 8], params[:, o + 8:o + 16], params[:, o + 16:o + 17]
    x = F.relu(torch.einsum("noi,nihw->nohw", w0, x) + b0[:, :, None, None])
    x = F.relu(torch.einsum("noi,nihw->nohw", w1, x) + b1[:, :, None, None])
    x = torch.einsum("noi,nihw->nohw", w2, x) + b2[:, :, None, None]
    return x[:, 0]


def add_layernorm_core(x: Tensor, y: Optional[Tensor], weight: Tensor, bias: Tensor, eps: float = 1e-5,
                       return_sum: bool = True):
    """Kernel-boundary form of K5: (x + y, LayerNorm(x + y)); y may be None."""
    s = x if y is None else x + y
    n = F.layer_norm(s, (s.shape[-1],), weight, bias, eps)
    return (s if (return_sum or y is None) else None), n


def msda_backward_core(value: Tensor, shapes: Tensor, level_start: Tensor, loc: Tensor, w: Tensor,
                       grad_out: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(grad_value, grad_loc, grad_w) of msda_core by autograd -- the quantity the reference's native
    backward (ms_deform_attn_cuda.cu:83-153) returns and its gradcheck (models/ops/test.py:62-80) verifies."""
    with torch.enable_grad():
        v, lo, ww = (t.detach().clone().requires_grad_(True) for t in (value, loc, w))
        out = msda_core(v, shapes, level_start, lo, ww)
        return torch.autograd.grad(out, (v, lo, ww), grad_out.reshape(out.shape))


def groupnorm_tokens_core(x: Tensor, weight: Tensor, bias: Tensor, groups: int, eps: float = 1e-5) -> Tensor:
    """Kernel-boundary form of K10: nn.GroupNorm (reference models/soc.py:107-125) applied to token-major
    x [N,S,C], i.e. to its '(n) c s' view."""
    return F.group_norm(x.transpose(1, 2), groups, weight, bias, eps).transpose(1, 2).contiguous()


def patch_merge_layernorm_core(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """Kernel-boundary form of K11: PatchMerging.forward without the reduction Linear
    (reference models/video_swin_transformer.py:296-311)."""
    H, W = x.shape[2], x.shape[3]
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def linear_core(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, add: Optional[Tensor] = None,
                relu: bool = False) -> Tensor:
    """Kernel-boundary form of K7: act((x + add) W^T + b), i.e. with_pos_embed + nn.Linear (+ ReLU) as
    in reference models/deformable_transformer.py:318-347 and models/voc.py:44-48,84-90."""
    y = F.linear(x if add is None else x + add, weight, bias)
    return F.relu(y) if relu else y


def linear_act_core(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: str = "none") -> Tensor:
    """Kernel-boundary form of K12: nn.Linear followed by nothing / ReLU / nn.GELU() (exact erf form), as in
    Mlp.forward of reference models/video_swin_transformer.py:24-37."""
    y = F.linear(x, weight, bias)
    return {"none": lambda t: t, "relu": F.relu, "gelu": F.gelu}[act](y)


def box_refine_core(delta: Tensor, ref: Tensor, valid_ratios: Optional[Tensor] = None):
    """Kernel-boundary form of K8: the iterative box refinement of reference
    models/deformable_transformer.py:369-381 plus the next layer's reference_points_input (:358-364)."""
    if ref.shape[-1] == 4:
        new = (delta + inverse_sigmoid(ref)).sigmoid()
    else:
        new = torch.cat([delta[..., :2] + inverse_sigmoid(ref), delta[..., 2:]], -1).sigmoid()
    ref_in = None
    if valid_ratios is not None:
        ref_in = new[:, :, None] * torch.cat([valid_ratios, valid_ratios], -1)[:, None]
    return new, ref_in


# ----------------------------------------------------------------------------- text
def build_text_encoder(sd: SD):
    """HF RobertaModel (third party, as in reference models/soc.py:104) loaded from text_encoder.*"""
    from transformers import RobertaConfig, RobertaModel
    cfg = RobertaConfig(vocab_size=sd["text_encoder.embeddings.word_embeddings.weight"].shape[0],
                        hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                        intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1,
                        layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    m = RobertaModel(cfg).eval()
    sub = {k[len("text_encoder."):]: v for k, v in sd.items() if k.startswith("text_encoder.")}
    missing, unexpected = m.load_state_dict(sub, strict=False)
    assert not unexpected and all(k.endswith(("position_ids", "token_type_ids")) for k in missing), (missing, unexpected)
    return m


def text_features(sd: SD, encoder, ids: Tensor, attn: Tensor):
    """reference SOC.forward_text models/soc.py:167-181 with pre-tokenised ids [B,L]"""
    enc = encoder(input_ids=ids, attention_mask=attn)
    words = layer_norm(sd, "txt_proj.layer_norm", linear(sd, "txt_proj.fc", enc.last_hidden_state.transpose(0, 1)), 1e-12)
    sent = layer_norm(sd, "txt_proj.layer_norm", linear(sd, "txt_proj.fc", enc.pooler_output), 1e-12)
    return words, attn.ne(1), sent


# ----------------------------------------------------------------------------- whole forward
@torch.no_grad()
def soc_forward(sd: SD, clip: Tensor, ids: Tensor, attn: Tensor, img_hw: Tuple[int, int],
                backbone: str = "video-swin-t", text_encoder=None, taps: Optional[dict] = None) -> Dict[str, Tensor]:
    """Eval forward for B=1 (reference SOC.forward models/soc.py:184-394, valid_indices=None).

    clip [T,3,H,W] (no padding: the pad mask is all False, as for every single-video batch built
    by misc.nested_tensor_from_videos_list).  Returns the level-0 dict (eval quirk).
    ``taps`` (optional dict) receives intermediate tensors for stage-boundary checks.
    """
    tap = (lambda k, v: taps.__setitem__(k, v)) if taps is not None else (lambda k, v: None)
    T, _, H, W = clip.shape
    B, Q = 1, sd["query_embed.weight"].shape[0]
    if text_encoder is None:
        text_encoder = build_text_encoder(sd)
    words, word_pad, sent = text_features(sd, text_encoder, ids, attn)  # [L,1,C], [1,L], [1,C]
    text_pos = sine_pos_1d(word_pad, 256).permute(2, 0, 1)  # [L,1,C]

    feats = swin_backbone(sd, clip, backbone)
    pad = torch.zeros(T, H, W, dtype=torch.bool)
    srcs, masks, poses, lang_last = [], [], [], None
    for l, f in enumerate(feats[1:]):
        tap(f"backbone{l + 1}", f)
        m = resize_mask(pad, f.shape[-2:])
        pe = sine_pos_2d(m)
        s = group_norm(sd, f"input_proj.{l}.1", conv(sd, f"input_proj.{l}.0", f), 32)
        n, c, h, w = s.shape
        s_seq = s.permute(0, 2, 3, 1).reshape(T * h * w, 1, c)  # (t h w) b c
        fused = mmf(sd, "vlf", s_seq, words, word_pad, text_pos)
        if l == 2:  # only langs[-1] is consumed (soc.py:304)
            pos_seq = pe.permute(0, 2, 3, 1).reshape(T * h * w, 1, c)
            lang_last = mmf(sd, "lvf", words, s_seq, m.reshape(1, -1), pos_seq)
        srcs.append(fused.view(T, h, w, c).permute(0, 3, 1, 2).contiguous())
        masks.append(m)
        poses.append(pe)
    tap("backbone0", feats[0])
    s = group_norm(sd, "input_proj.3.1", conv(sd, "input_proj.3.0", feats[3], stride=2, padding=1), 32)
    m = resize_mask(pad, s.shape[-2:])
    pe = sine_pos_2d(m)
    n, c, h, w = s.shape
    fused = mmf(sd, "vlf", s.permute(0, 2, 3, 1).reshape(T * h * w, 1, c), words, word_pad, text_pos)
    srcs.append(fused.view(T, h, w, c).permute(0, 3, 1, 2).contiguous())
    masks.append(m)
    poses.append(pe)
    for l, s_ in enumerate(srcs):
        tap(f"src{l}", s_)

    hs, memory, init_ref, inter_refs = deformable_transformer(sd, srcs, masks, poses, sd["query_embed.weight"])
    tap("hs", hs)
    tap("inter_refs", inter_refs)
    for l, m_ in enumerate(memory):
        tap(f"memory{l}", m_)

    # vl-loss text feature: mean of the un-padded word rows of lvf (soc.py:298-310)
    text_feat = lang_last.transpose(0, 1)[0][~word_pad[0]].mean(0, keepdim=True)

    voc_hs = voc(sd, hs[-1].view(T, B, Q, -1), sent)  # [B,Q,C]
    tap("voc_hs", voc_hs)
    hs0 = hs[0] + voc_hs  # level 0 only is returned (T,Q,C with B=1)

    cls = linear(sd, "class_embed.0", hs0)
    box = mlp_relu(sd, "bbox_embed.0", hs0, 3).clone()
    box[..., :2] = box[..., :2] + inverse_sigmoid(init_ref)
    box = box.sigmoid()

    mem = [feats[0]] + memory  # soc.py:349-352
    fpn = fpn_spatial_decoder(sd, mem[-1], mem[:-1][::-1])
    tap("fpn", fpn)
    params = mlp_relu(sd, "controller", hs0, 3).reshape(T * Q, -1)
    refs0 = inter_refs[0][..., :2].reshape(T * Q, 2)
    tap("mask_params", params)
    masks_out = dynamic_mask_core(fpn, params, refs0, img_hw)
    hm, wm = masks_out.shape[-2:]
    return {
        "pred_masks": masks_out.view(T, 1, Q, hm, wm),
        "pred_cls": cls.view(T, 1, Q, -1),
        "pred_boxes": box.view(T, 1, Q, 4),
        "pred_logit": voc_hs,
        "text_sentence_feature": text_feat,
        "aux_outputs": [],
    }


def select_query(out: Dict[str, Tensor]) -> Tuple[int, Tensor]:
    """reference infer_refytb.py:216-227: best query by mean sigmoid score -> (index, masks [T,h,w])"""
    scores = out["pred_cls"][:, 0].sigmoid().mean(0).max(-1)[0]
    qi = int(scores.argmax())
    return qi, out["pred_masks"][:, 0, qi]
