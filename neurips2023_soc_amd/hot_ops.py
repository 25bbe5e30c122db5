"""Host-side entry points of the HIP kernels K1-K19 (torch tensors in, torch tensors out).

Every function requires CUDA(=HIP) tensors and calls straight into libsoc_hip.so through the C
ABI of include/soc_hip.h on torch's current stream.  There is deliberately no fallback: CPU
tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------
# measurement hooks (per-kernel event timing, recording of a forward's launches): op_profile.py; the names bench.py and the tests
# use are kept here
from . import op_profile  # noqa: E402
from .op_profile import profile_begin, profile_end  # noqa: E402,F401
from .op_profile import timed as _timed  # noqa: E402


def record_window_attention_calls(on: bool):
    """K1: the recorded tuples are the arguments of window_attention3d (op_profile.record_calls)."""
    return op_profile.record_calls("k1", on)


def record_linear_split_calls(on: bool):
    """K20: the recorded dicts are keyword arguments of linear_split."""
    return op_profile.record_calls("k20", on)


def record_ws_linear_calls(on: bool):
    """K13 / K13b: the recorded dicts are keyword arguments of ws_linear."""
    return op_profile.record_calls("k13", on)


def record_mlp_split_calls(on: bool):
    """K23: the recorded dicts are keyword arguments of mlp_split."""
    return op_profile.record_calls("k23", on)


def record_xs_linear_calls(on: bool):
    """K24: the recorded dicts are keyword arguments of xs_linear."""
    return op_profile.record_calls("k24", on)


def _need_gpu(*ts: Tensor) -> None:
    cur = None
    for x in ts:
        if x is None:
            continue
        if not x.is_cuda:
            raise _lib.SocHipError(
                "SOC hot ops run only on an MI355X (HIP) device tensor; got a CPU tensor. "
                "There is no CPU fallback in the product path.")
        if cur is None:
            cur = torch.cuda.current_device()
        if x.device.index != cur:
            # the kernels are launched on the CURRENT device's stream: a tensor of another device would be
            # dereferenced on the wrong GPU
            raise _lib.SocHipError(f"tensor on cuda:{x.device.index} but the current device is cuda:{cur}; "
                                   "wrap the call in torch.cuda.device(tensor.device)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream      # current device == every argument's device (_need_gpu)


def _plan_stream():
    """Stream argument of the plan queries (its CU mask sizes the plan); None = the null stream where there is no GPU."""
    return _stream() if torch.cuda.is_available() else None


def stream_cus() -> int:
    """CUs a launch on the current stream may use (the device's count, or the stream's CU mask: soc_stream_cus)."""
    return int(_lib.load().soc_stream_cus(_stream()))


def _f32c(x: Tensor) -> Tensor:
    if x.dtype != torch.float32:
        raise _lib.SocHipError(f"expected float32, got {x.dtype}")
    return x if x.is_contiguous() else x.contiguous()


def msda_forward(value: Tensor, spatial_shapes: Tensor, level_start_index: Tensor,
                 sampling_loc: Tensor, attn_weight: Tensor) -> Tensor:
    """K2.  value [N,S,M,D], shapes [L,2] i64, lsi [L] i64, loc [N,Lq,M,L,P,2], w [N,Lq,M,L,P]
    -> [N,Lq,M*D].  float32 or float64 (reference AT_DISPATCH_FLOATING_TYPES)."""
    _need_gpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    lib = _lib.load()
    for name, x in (("value", value), ("spatial_shapes", spatial_shapes),
                    ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                    ("attn_weight", attn_weight)):
        if not x.is_contiguous():  # reference ms_deform_attn_cuda.cu:28-32
            raise RuntimeError(f"{name} tensor has to be contiguous")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    if value.dtype == torch.float32:
        fn = lib.soc_msda_fwd_f32
    elif value.dtype == torch.float64:
        fn = lib.soc_msda_fwd_f64
    else:
        raise RuntimeError("ms_deform_attn_forward supports float32/float64 only")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("value / sampling_loc / attn_weight dtypes differ")
    # algorithmic bytes: value + loc + weights read once, out written once
    work = (value.numel() + sampling_loc.numel() + attn_weight.numel() + out.numel()) * value.element_size()
    with _timed("msda_fwd", work):
        code = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                  sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(), N, S, M, D, L, Lq, P,
                  _stream())
    _lib.check(code, "soc_msda_fwd")
    return out


def msda_backward(value: Tensor, spatial_shapes: Tensor, level_start_index: Tensor, sampling_loc: Tensor,
                  attn_weight: Tensor, grad_output: Tensor):
    """K2 backward.  Same tensors as msda_forward plus grad_output [N,Lq,M*D] ->
    (grad_value [N,S,M,D], grad_sampling_loc [N,Lq,M,L,P,2], grad_attn_weight [N,Lq,M,L,P])."""
    _need_gpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    lib = _lib.load()
    if value.dtype == torch.float32:
        fn = lib.soc_msda_bwd_f32
    elif value.dtype == torch.float64:
        fn = lib.soc_msda_bwd_f64
    else:
        raise RuntimeError("ms_deform_attn_backward supports float32/float64 only")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    tensors = [value, sampling_loc, attn_weight, grad_output]
    if any(x.dtype != value.dtype for x in tensors):
        raise RuntimeError("value / sampling_loc / attn_weight / grad_output dtypes differ")
    value, sampling_loc, attn_weight, grad_output = (x.contiguous() for x in tensors)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    with _timed("msda_bwd", (2 * value.numel() + 2 * sampling_loc.numel() + 2 * attn_weight.numel()
                             + grad_output.numel()) * value.element_size()):
        code = fn(value.data_ptr(), spatial_shapes.contiguous().data_ptr(), level_start_index.contiguous().data_ptr(),
                  sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(), gv.data_ptr(),
                  gl.data_ptr(), ga.data_ptr(), N, S, M, D, L, Lq, P, _stream())
    _lib.check(code, "soc_msda_bwd")
    return gv, gl, ga


def msda_fused_forward(value: Tensor, spatial_shapes: Tensor, level_start_index: Tensor,
                       reference_points: Tensor, offsets: Tensor, logits: Tensor,
                       pad_mask: Optional[Tensor] = None, any_pad: Optional[Tensor] = None) -> Tensor:
    """K2 fused form.  value [N,S,M,32] (un-masked), reference_points [N,Lq,4,2|4], offsets
    [N,Lq,M,4,4,2] raw, logits [N,Lq,M,16] raw, pad_mask [N,S] bool/uint8 + any_pad int32[1]
    (both or neither) -> [N,Lq,M*32]."""
    _need_gpu(value, spatial_shapes, level_start_index, reference_points, offsets, logits, pad_mask, any_pad)
    lib = _lib.load()
    value, reference_points, offsets, logits = (_f32c(x) for x in (value, reference_points, offsets, logits))
    N, S, M, D = value.shape
    Lq, L, rd = reference_points.shape[1], reference_points.shape[2], reference_points.shape[3]
    P = offsets.shape[4]
    out = torch.empty((N, Lq, M * D), dtype=torch.float32, device=value.device)
    pm_ptr = ap_ptr = None
    if pad_mask is not None:
        pm = _as_u8(pad_mask)
        if any_pad is None or any_pad.dtype != torch.int32:
            raise _lib.SocHipError("msda_fused_forward: pad_mask needs an int32 any_pad flag tensor")
        pm_ptr, ap_ptr = pm.data_ptr(), any_pad.data_ptr()
    work = (value.numel() + offsets.numel() + logits.numel() + out.numel()) * 4
    with _timed("msda_fwd", work):
        code = lib.soc_msda_fused_fwd_f32(value.data_ptr(), pm_ptr, ap_ptr, spatial_shapes.data_ptr(),
                                          level_start_index.data_ptr(), reference_points.data_ptr(), rd,
                                          offsets.data_ptr(), logits.data_ptr(), out.data_ptr(),
                                          N, S, M, D, L, Lq, P, _stream())
    _lib.check(code, "soc_msda_fused_fwd_f32")
    return out


# K16 fuses 2-3 dependent launches of the query chain into one 16-wave workgroup per row: fewer launches and less
# latency for a chain that owns the GPU, but more CU time than K7's small workgroups.  A chain that runs beside another
# clip's chip-filling kernels (graph_runner.PipelinedClipGraph's tail) is charged for CU time, not for launches, and
# switches it off (measured 8.70 vs 8.76 ms per clip).
row_chain_fusion = True

DECODER_XATTN_MAX_ROWS = 2048     # one workgroup per (frame, query) row re-reads 896 KB of weights: few rows only


def decoder_cross_attn_supported(tgt: Tensor, cross_attn, memory: Tensor, reference_points: Tensor) -> bool:
    """True when K15 covers this call (cross_attn: an MSDeformAttn module)."""
    return (tgt.is_cuda and tgt.dtype == torch.float32 and memory.dtype == torch.float32
            and cross_attn.d_model == 256 and cross_attn.n_heads == 8 and cross_attn.n_levels == 4
            and cross_attn.n_points == 4 and reference_points.shape[-1] in (2, 4)
            and tgt.shape[0] * tgt.shape[1] <= DECODER_XATTN_MAX_ROWS and memory.shape[1] * 256 < (1 << 31))


def decoder_cross_attn(tgt: Tensor, query_pos: Tensor, reference_points: Tensor, memory: Tensor,
                       spatial_shapes: Tensor, level_start_index: Tensor, cross_attn, norm,
                       pad_mask: Optional[Tensor] = None, any_pad: Optional[Tensor] = None) -> Tensor:
    """K15.  norm(tgt + cross_attn(tgt + query_pos, reference_points, memory)) for a deformable-decoder layer in one
    launch: tgt [N,Lq,256], query_pos [N,Lq,256] or [Lq,256], reference_points [N,Lq,4,2|4], memory [N,S,256];
    cross_attn: MSDeformAttn (sampling_offsets, attention_weights, value_proj, output_proj), norm: nn.LayerNorm."""
    _need_gpu(tgt, query_pos, reference_points, memory, spatial_shapes, level_start_index, pad_mask, any_pad)
    lib = _lib.load()
    tgt, reference_points, memory = _f32c(tgt), _f32c(reference_points), _f32c(memory)
    N, Lq, C = tgt.shape
    if query_pos.dim() == 3 and query_pos.stride(0) == 0:
        query_pos = query_pos[0]                      # an expanded [Lq,256] embedding
    query_pos = _f32c(query_pos)
    per_frame = 1 if query_pos.dim() == 3 else 0
    if query_pos.shape[-2:] != (Lq, C) or (per_frame and query_pos.shape[0] != N):
        raise _lib.SocHipError(f"decoder_cross_attn: query_pos {tuple(query_pos.shape)} does not match tgt {tuple(tgt.shape)}")
    S = memory.shape[1]
    pm_ptr = ap_ptr = None
    if pad_mask is not None:
        pm = _as_u8(pad_mask)
        if any_pad is None or any_pad.dtype != torch.int32:
            raise _lib.SocHipError("decoder_cross_attn: pad_mask needs an int32 any_pad flag tensor")
        pm_ptr, ap_ptr = pm.data_ptr(), any_pad.data_ptr()
    ca = cross_attn
    ws = [_f32c(t) for t in (ca.sampling_offsets.weight, ca.sampling_offsets.bias, ca.attention_weights.weight,
                             ca.attention_weights.bias, ca.value_proj.weight, ca.value_proj.bias,
                             ca.output_proj.weight, ca.output_proj.bias, norm.weight, norm.bias)]
    out = torch.empty_like(tgt)
    with _timed("decoder_cross_attn", (tgt.numel() * 2 + sum(w.numel() for w in ws) * N * Lq) * 4):
        code = lib.soc_decoder_cross_attn_f32(
            tgt.data_ptr(), query_pos.data_ptr(), per_frame, reference_points.data_ptr(), reference_points.shape[-1],
            memory.data_ptr(), pm_ptr, ap_ptr, spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            *[w.data_ptr() for w in ws], float(norm.eps), out.data_ptr(), N, Lq, S, C, ca.n_heads, ca.n_levels,
            ca.n_points, _stream())
    _lib.check(code, "soc_decoder_cross_attn_f32")
    return out


ROW_MLP_MAX_ROWS = 2048     # one workgroup per row re-reads the layers' weights: few rows only


def row_mlp_supported(x: Tensor, weights: Sequence[Tensor], has_ln: bool = False) -> bool:
    """True when K16 covers `x -> Linear(+ReLU) ... -> Linear` over these weight matrices."""
    if not row_chain_fusion:
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and 1 <= len(weights) <= 3 and x.shape[-1] == 256
            and 0 < x.numel() // 256 <= ROW_MLP_MAX_ROWS):
        return False
    if any(w.shape[1] != 256 or w.dtype != torch.float32 for w in weights) or any(w.shape[0] != 256 for w in weights[:-1]):
        return False
    n_out = weights[-1].shape[0]
    return n_out == 256 if has_ln else n_out <= 256


def row_mlp(x: Tensor, layers: Sequence[Tuple[Tensor, Optional[Tensor]]], add: Optional[Tensor] = None,
            residual: Optional[Tensor] = None, ln: Optional[Tuple[Tensor, Tensor, float]] = None) -> Tensor:
    """K16.  layers = [(weight [N,256], bias | None), ...] (ReLU between layers, none after the last);
    out = y (+ residual), or LayerNorm(residual + y) with ln = (gamma, beta, eps).  x [..., 256]; `add` (a positional
    embedding added to x) must have x's shape or be [rows_mod, 256] broadcast over the leading rows."""
    _need_gpu(x, add, residual, *[t for wb in layers for t in wb])
    lib = _lib.load()
    x = _f32c(x)
    M = x.numel() // 256
    ws = [_f32c(w) for w, _ in layers]
    bs = [None if b is None else _f32c(b) for _, b in layers]
    n_out = ws[-1].shape[0]
    add_ptr, add_div, add_mod = None, 1, 1
    if add is not None:
        add = _f32c(add)
        add_ptr, add_mod = add.data_ptr(), add.numel() // 256
        if add_mod != M and x.shape[-2] != add_mod:
            raise _lib.SocHipError(f"row_mlp: add {tuple(add.shape)} does not broadcast over x {tuple(x.shape)}")
    res_ptr = None
    if residual is not None:
        residual = _f32c(residual)
        if residual.numel() != M * n_out:
            raise _lib.SocHipError("row_mlp: residual shape does not match the output")
        res_ptr = residual.data_ptr()
    g_ptr = b_ptr = None
    eps = 0.0
    if ln is not None:
        gamma, beta, eps = _f32c(ln[0]), _f32c(ln[1]), float(ln[2])
        g_ptr, b_ptr = gamma.data_ptr(), beta.data_ptr()
    out = torch.empty(x.shape[:-1] + (n_out,), dtype=torch.float32, device=x.device)
    wp = (C.c_void_p * 3)(*[w.data_ptr() for w in ws], *([None] * (3 - len(ws))))
    has_bias = any(b is not None for b in bs)
    bp = (C.c_void_p * 3)(*[None if b is None else b.data_ptr() for b in bs], *([None] * (3 - len(bs)))) if has_bias else None
    with _timed("row_mlp", (x.numel() + out.numel() + M * sum(w.numel() for w in ws)) * 4):
        code = lib.soc_row_mlp_f32(x.data_ptr(), add_ptr, add_div, add_mod, len(ws), wp, bp, n_out, res_ptr, g_ptr, b_ptr,
                                   eps, out.data_ptr(), M, 256, _stream())
    _lib.check(code, "soc_row_mlp_f32")
    return out


def groupnorm_nchw_supported(x: Tensor, groups: int) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        return False
    N, Cc, H, W = x.shape
    return Cc % groups == 0 and (H * W) % 4 == 0 and (Cc // groups) * H * W <= 32768


def groupnorm_nchw(x: Tensor, conv_bias: Optional[Tensor], weight: Tensor, bias: Tensor, groups: int, eps: float,
                   relu: bool = True) -> Tensor:
    """K17.  relu(GroupNorm(x + conv_bias[c])) on an NCHW map (x is overwritten when it owns its memory: it is the
    bias-free convolution output)."""
    _need_gpu(x, conv_bias, weight, bias)
    lib = _lib.load()
    x = _f32c(x)
    N, Cc, H, W = x.shape
    with _timed("fpn_elementwise", x.numel() * 8):
        code = lib.soc_groupnorm_nchw_f32(x.data_ptr(), None if conv_bias is None else _f32c(conv_bias).data_ptr(),
                                          _f32c(weight).data_ptr(), _f32c(bias).data_ptr(), x.data_ptr(), N, Cc, H * W,
                                          groups, float(eps), int(relu), _stream())
    _lib.check(code, "soc_groupnorm_nchw_f32")
    return x


def upsample_add_nchw(lateral: Tensor, conv_bias: Optional[Tensor], prev: Tensor) -> Tensor:
    """K18.  lateral + conv_bias[c] + F.interpolate(prev, size=lateral.shape[-2:], mode="nearest") (lateral is
    overwritten: it is the bias-free adapter output)."""
    _need_gpu(lateral, conv_bias, prev)
    lib = _lib.load()
    lateral, prev = _f32c(lateral), _f32c(prev)
    N, Cc, H, W = lateral.shape
    if prev.shape[:2] != (N, Cc):
        raise _lib.SocHipError(f"upsample_add_nchw: {tuple(prev.shape)} vs {tuple(lateral.shape)}")
    with _timed("fpn_elementwise", lateral.numel() * 8 + prev.numel() * 4):
        code = lib.soc_upsample_add_nchw_f32(lateral.data_ptr(), None if conv_bias is None else _f32c(conv_bias).data_ptr(),
                                             prev.data_ptr(), lateral.data_ptr(), N, Cc, H, W, prev.shape[2],
                                             prev.shape[3], _stream())
    _lib.check(code, "soc_upsample_add_nchw_f32")
    return lateral


def conv3x3_tokens(x: Tensor, hw: Tuple[int, int], w_taps: Tensor, bias: Optional[Tensor], out_nchw: bool = False,
                   relu: bool = False) -> Tensor:
    """K19.  3x3 / pad 1 convolution of a token-major map.  x [N, H*W, Cin] whose frames may be strided (a level slice of
    the encoder memory: x.stride() == (frame_stride, Cin, 1)); w_taps [Cout, 9*Cin] = conv.weight.permute(0, 2, 3, 1);
    -> [N, H*W, Cout] (token-major) or [N, Cout, H, W]."""
    _need_gpu(x, w_taps, bias)
    lib = _lib.load()
    N, S, Cin = x.shape
    H, W = hw
    if x.dtype != torch.float32 or S != H * W or x.stride(2) != 1 or x.stride(1) != Cin or (N > 1 and x.stride(0) < S * Cin):
        raise _lib.SocHipError(f"conv3x3_tokens: unsupported input layout {tuple(x.shape)} / {x.stride()}")
    w_taps = _f32c(w_taps)
    Cout = w_taps.shape[0]
    if w_taps.shape[1] != 9 * Cin:
        raise _lib.SocHipError("conv3x3_tokens: w_taps must be [Cout, 9*Cin]")
    out = torch.empty((N, Cout, H, W) if out_nchw else (N, S, Cout), dtype=torch.float32, device=x.device)
    with _timed("conv3x3_tokens", (x.numel() + w_taps.numel() + out.numel()) * 4):
        code = lib.soc_conv3x3_tokens_f32(x.data_ptr(), x.stride(0) if N > 1 else S * Cin, w_taps.data_ptr(),
                                          None if bias is None else _f32c(bias).data_ptr(), out.data_ptr(), N, H, W, Cin,
                                          Cout, int(out_nchw), int(relu), _stream())
    _lib.check(code, "soc_conv3x3_tokens_f32")
    return out


def upsample_add_tokens(lateral: Tensor, bias: Optional[Tensor], prev: Tensor, hw: Tuple[int, int],
                        hw_prev: Tuple[int, int]) -> Tensor:
    """K18, token-major form.  lateral [N, H*W, C] + bias + nearest-up-sampled prev [N, Hp*Wp, C] (lateral is overwritten
    when it owns its memory)."""
    _need_gpu(lateral, bias, prev)
    lib = _lib.load()
    lateral, prev = _f32c(lateral), _f32c(prev)
    N, S, Cc = lateral.shape
    (H, W), (Hp, Wp) = hw, hw_prev
    if S != H * W or prev.shape != (N, Hp * Wp, Cc):
        raise _lib.SocHipError(f"upsample_add_tokens: {tuple(lateral.shape)} / {tuple(prev.shape)} vs {hw} / {hw_prev}")
    with _timed("fpn_elementwise", lateral.numel() * 8 + prev.numel() * 4):
        code = lib.soc_upsample_add_tokens_f32(lateral.data_ptr(), None if bias is None else _f32c(bias).data_ptr(),
                                               prev.data_ptr(), lateral.data_ptr(), N, Cc, H, W, Hp, Wp, _stream())
    _lib.check(code, "soc_upsample_add_tokens_f32")
    return lateral


def clamp_window(size: Sequence[int], window: Sequence[int], shift: Sequence[int]):
    """get_window_size of the reference (models/video_swin_transformer.py:71-84)."""
    w, s = list(window), list(shift)
    for i in range(3):
        if size[i] <= window[i]:
            w[i], s[i] = size[i], 0
    return tuple(w), tuple(s)


def _k1_form() -> int:
    """The `split_arith` launch argument of K1: 0 = f32-input MFMA, 1 = the bf16-split STREAMING form (round 6: 32-query tiles,
    key chunks of 32, softmax between the MFMAs), 2 = round 3's split form (SOC_K1_FORM=r3: kept for A/B runs and its tests)."""
    if not k1_split_enabled():
        return 0
    return 2 if _os.environ.get("SOC_K1_FORM", "") == "r3" else 1


def window_attention3d(qkv: Tensor, qkv_bias: Tensor, bias_table: Tensor, n_heads: int,
                       window: Sequence[int], shift: Sequence[int]) -> Tensor:
    """K1.  qkv [B,D,H,W,3C] (token layout, un-padded) -> attention output [B,D,H,W,C].
    ``window``/``shift`` are the module's nominal values; clamping happens here."""
    _need_gpu(qkv, qkv_bias, bias_table)
    lib = _lib.load()
    qkv, qkv_bias, bias_table = _f32c(qkv), _f32c(qkv_bias), _f32c(bias_table)
    _rec = op_profile.recording("k1")
    if _rec is not None:
        _rec.append((qkv, qkv_bias, bias_table, n_heads, tuple(window), tuple(shift)))
    B, D, H, W, C3 = qkv.shape
    C = C3 // 3
    win, sh = clamp_window((D, H, W), window, shift)
    out = torch.empty((B, D, H, W, C), dtype=torch.float32, device=qkv.device)
    # algorithmic FLOPs: QK^T + PV = 4 * N^2 * head_dim per (window, head)
    n_win = B * -(-D // win[0]) * -(-H // win[1]) * -(-W // win[2])
    n_tok = win[0] * win[1] * win[2]
    with _timed("win_attn3d", 4.0 * n_tok * n_tok * (C // n_heads) * n_win * n_heads):
        code = lib.soc_win_attn3d_f32(qkv.data_ptr(), qkv_bias.data_ptr(), bias_table.data_ptr(),
                                      out.data_ptr(), B, D, H, W, C, n_heads, *win, *sh, *window,
                                      _k1_form(), _stream())
    _lib.check(code, "soc_win_attn3d_f32")
    return out


_ws_cache = {}


def mha_core(q: Tensor, k: Tensor, v: Tensor, n_heads: int,
             key_padding_mask: Optional[Tensor] = None, batch_first: bool = False,
             attn_mask: Optional[Tensor] = None) -> Tensor:
    """K3.  q [Lq,B,E], k/v [Lk,B,E] projected, key_padding_mask [B,Lk] bool -> [Lq,B,E]
    (batch_first: q [B,Lq,E], k/v [B,Lk,E] -> [B,Lq,E]).  attn_mask: float, added to the scaled logits,
    [Lq,Lk], [B,Lq,Lk] or torch's [B*n_heads,Lq,Lk]."""
    _need_gpu(q, k, v, key_padding_mask, attn_mask)
    lib = _lib.load()
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    if batch_first:
        B, Lq, E = q.shape
        Lk = k.shape[1]
    else:
        Lq, B, E = q.shape
        Lk = k.shape[0]
    hd = E // n_heads
    out = torch.empty_like(q)
    kpm_ptr = None
    if key_padding_mask is not None:
        kpm = _as_u8(key_padding_mask)
        kpm_ptr = kpm.data_ptr()
    am_ptr, am_heads = None, 1
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            raise _lib.SocHipError("mha_core: boolean attn_mask is not supported; pass an additive float mask")
        am = _f32c(attn_mask if attn_mask.dim() == 3 else attn_mask[None].expand(B, -1, -1))
        if am.shape[-2:] != (Lq, Lk) or am.shape[0] not in (B, B * n_heads):
            raise _lib.SocHipError(f"mha_core: attn_mask shape {tuple(attn_mask.shape)} does not fit "
                                   f"[{B} or {B * n_heads}, {Lq}, {Lk}]")
        am_ptr, am_heads = am.data_ptr(), (n_heads if am.shape[0] == B * n_heads and n_heads > 1 else 1)
    need = lib.soc_xattn_workspace_bytes(Lq, Lk, B, n_heads, hd)
    ws_ptr = None
    if need:
        key = (q.device.index, torch.cuda.current_stream().cuda_stream)
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=q.device)
            _ws_cache[key] = ws
        ws_ptr = ws.data_ptr()
    with _timed("xattn", (2 * q.numel() + k.numel() + v.numel()) * 4):
        code = lib.soc_xattn_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), kpm_ptr, am_ptr, am_heads, out.data_ptr(),
                                 Lq, Lk, B, n_heads, hd, int(bool(batch_first)), ws_ptr, need, _stream())
    _lib.check(code, "soc_xattn_f32")
    return out


def dynamic_mask(feats: Tensor, params: Tensor, refs: Tensor, img_hw: Sequence[float],
                 stride: int = 4) -> Tensor:
    """K4.  feats [T,C,h,w], params [T*Q,169], refs [T*Q,2], img_hw=(H_img,W_img) -> [T*Q,h,w]."""
    _need_gpu(feats, params, refs)
    lib = _lib.load()
    feats, params, refs = _f32c(feats), _f32c(params), _f32c(refs)
    T, Cm, h, w = feats.shape
    TQ = params.shape[0]
    Q = TQ // max(T, 1)
    if params.shape[1] != (Cm + 2) * 8 + 64 + 8 + 8 + 8 + 1:
        raise _lib.SocHipError(f"unexpected dynamic-parameter count {params.shape[1]}")
    out = torch.empty((TQ, h, w), dtype=torch.float32, device=feats.device)
    with _timed("dyn_mask", (feats.numel() + params.numel() + refs.numel() + out.numel()) * 4):
        code = lib.soc_dyn_mask_f32(feats.data_ptr(), params.data_ptr(), refs.data_ptr(), out.data_ptr(),
                                    T, Q, Cm, h, w, float(img_hw[0]), float(img_hw[1]), stride, _stream())
    _lib.check(code, "soc_dyn_mask_f32")
    return out


def add_layernorm(x: Tensor, y: Optional[Tensor], weight: Tensor, bias: Tensor, eps: float = 1e-5,
                  return_sum: bool = True):
    """K5.  (x + y, LayerNorm(x + y)) over the last dim in one pass; y may be None (plain LN).
    Returns (sum, norm) -- sum is x itself when y is None."""
    _need_gpu(x, y, weight, bias)
    lib = _lib.load()
    x = _f32c(x)
    C = x.shape[-1]
    rows = x.numel() // C
    out_norm = torch.empty_like(x)
    if y is None:
        out_sum, y_ptr, sum_ptr = x, None, None
    else:
        y = _f32c(y)
        if y.shape != x.shape:
            raise _lib.SocHipError("add_layernorm: x and y shapes differ")
        out_sum = torch.empty_like(x) if return_sum else None
        y_ptr, sum_ptr = y.data_ptr(), (out_sum.data_ptr() if return_sum else None)
    n_out = 2 if (y is not None and return_sum) else 1
    with _timed("add_layernorm", (x.numel() * (1 + (y is not None)) + n_out * x.numel()) * 4):
        code = lib.soc_add_layernorm_f32(x.data_ptr(), y_ptr, _f32c(weight).data_ptr(), _f32c(bias).data_ptr(),
                                         sum_ptr, out_norm.data_ptr(), rows, C, float(eps), _stream())
    _lib.check(code, "soc_add_layernorm_f32")
    return out_sum, out_norm


def upsample_threshold(mask_logits: Tensor, size: Sequence[int], threshold_logit: float = 0.0) -> Tensor:
    """K6.  [T,h,w] logits -> bool [T,H0,W0]: bilinear (align_corners=False) up-sampling fused with
    the sigmoid > 0.5 threshold (as logit > 0)."""
    _need_gpu(mask_logits)
    lib = _lib.load()
    x = _f32c(mask_logits)
    T, h, w = x.shape
    H0, W0 = int(size[0]), int(size[1])
    out = torch.empty((T, H0, W0), dtype=torch.uint8, device=x.device)
    with _timed("upsample_threshold", x.numel() * 4 + out.numel()):
        code = lib.soc_upsample_threshold_u8(x.data_ptr(), out.data_ptr(), T, h, w, H0, W0, float(threshold_logit),
                                             _stream())
    _lib.check(code, "soc_upsample_threshold_u8")
    return out.view(torch.bool)


SMALL_LINEAR_MAX_ROWS = 1024


def _broadcast_rows(add: Tensor, lead: Sequence[int], K: int):
    """How a positional term broadcasts over the flattened rows of x: returns (base [R,K] contiguous,
    div, mod) with row m of x + add == x[m] + base[(m // div) % mod], or None if it is not of that form."""
    if add.dim() != len(lead) + 1 or add.shape[-1] != K or add.stride(-1) != 1:
        return None
    if tuple(add.shape[:-1]) != tuple(lead):
        return None
    live = [d for d in range(len(lead)) if lead[d] > 1 and add.stride(d) != 0]
    if len(live) > 1:
        # every leading dim varies: must be a plain contiguous [M,K]
        return (add.reshape(-1, K), 1, int(add.numel() // K)) if add.is_contiguous() else None
    if not live:
        return add.as_strided((1, K), (K, 1)), 1, 1
    d = live[0]
    if add.stride(d) != K:
        return None
    inner = 1
    for e in lead[d + 1:]:
        inner *= int(e)
    return add.as_strided((lead[d], K), (K, 1)), inner, int(lead[d])


def _as_u8(mask: Tensor) -> Tensor:
    """bool / uint8 mask as contiguous uint8 without a conversion kernel (bool is stored as 0/1 bytes)."""
    mask = mask.contiguous()
    if mask.dtype == torch.bool:
        return mask.view(torch.uint8)
    return mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)


def linear_small_multi(x: Tensor, layers: Sequence[Tuple[Tensor, Optional[Tensor], bool]],
                       add: Optional[Tensor] = None, relu: bool = False) -> List[Tensor]:
    """K7.  Several linear layers over the same few-row input in one launch.
    layers = [(weight [N_i,K], bias [N_i] | None, use_add), ...] (at most 4); returns
    [act((x + add if use_add else x) @ weight.T + bias) for each layer].  `add` must have x's shape,
    possibly as an expanded (stride-0) view."""
    _need_gpu(x, *(w for w, _, _ in layers))
    lib = _lib.load()
    x = _f32c(x)
    K = x.shape[-1]
    M = x.numel() // K
    add_ptr, div, mod = None, 1, 1
    if add is not None and any(u for _, _, u in layers):
        form = _broadcast_rows(add, x.shape[:-1], K) if add.dtype == torch.float32 else None
        if form is None:
            base, div, mod = _f32c(add.expand_as(x)).reshape(-1, K), 1, max(M, 1)
        else:
            base, div, mod = form
        add_ptr = base.data_ptr()
    n = len(layers)
    ws = [_f32c(w) for w, _, _ in layers]
    bs = [None if b is None else _f32c(b) for _, b, _ in layers]
    outs = [torch.empty(*x.shape[:-1], w.shape[0], dtype=torch.float32, device=x.device) for w in ws]
    vp = C.c_void_p * n
    w_arr = vp(*(w.data_ptr() for w in ws))
    b_arr = vp(*(None if b is None else b.data_ptr() for b in bs))
    o_arr = vp(*(o.data_ptr() for o in outs))
    n_arr = (C.c_int * n)(*(w.shape[0] for w in ws))
    u_arr = (C.c_int * n)(*(int(bool(u)) for _, _, u in layers))
    with _timed("linear_small", (M * K + sum(w.numel() + M * w.shape[0] for w in ws)) * 4):
        code = lib.soc_linear_small_multi_f32(x.data_ptr(), add_ptr, div, mod, n, w_arr, b_arr, o_arr, n_arr, u_arr,
                                              M, K, 2 if relu == "gelu" else int(bool(relu)), _stream())
    _lib.check(code, "soc_linear_small_multi_f32")
    return outs


def linear_small(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, add: Optional[Tensor] = None,
                 relu: bool = False) -> Tensor:
    """K7.  act((x [+ add]) @ weight.T + bias) for few rows (x.numel() / K <= SMALL_LINEAR_MAX_ROWS is
    what callers use it for)."""
    return linear_small_multi(x, [(weight, bias, add is not None)], add, relu)[0]


def box_refine(delta: Tensor, ref: Tensor, valid_ratios: Optional[Tensor] = None):
    """K8.  delta [N,Q,4], ref [N,Q,2|4], valid_ratios [N,L,2] | None ->
    (new_ref [N,Q,4] = sigmoid(delta + inverse_sigmoid(ref)) (only x, y refined when ref has 2 columns),
     ref_in [N,Q,L,4] = new_ref[:, :, None] * cat(valid_ratios, valid_ratios)[:, None], or None)."""
    _need_gpu(delta, ref, valid_ratios)
    lib = _lib.load()
    delta, ref = _f32c(delta), _f32c(ref)
    N, Q = delta.shape[:2]
    new_ref = torch.empty(N, Q, 4, dtype=torch.float32, device=delta.device)
    ref_in, vr_ptr, in_ptr, L = None, None, None, 0
    if valid_ratios is not None:
        vr = _f32c(valid_ratios)
        L = vr.shape[1]
        ref_in = torch.empty(N, Q, L, 4, dtype=torch.float32, device=delta.device)
        vr_ptr, in_ptr = vr.data_ptr(), ref_in.data_ptr()
    with _timed("box_refine", (delta.numel() + ref.numel() + new_ref.numel() * (1 + L)) * 4):
        code = lib.soc_box_refine_f32(delta.data_ptr(), ref.data_ptr(), ref.shape[-1], vr_ptr, new_ref.data_ptr(),
                                      in_ptr, N, Q, L, _stream())
    _lib.check(code, "soc_box_refine_f32")
    return new_ref, ref_in


def select_pack(pred_cls: Tensor, pred_masks: Tensor, records: Tensor) -> None:
    """K26.  pred_cls [T,B,Q,K] (any strides over t / b / q, K contiguous), pred_masks [T,B,Q,h,w] -> records [B, 1 + T*Q + T*h*w]
    float32 (rows may be strided): per clip the selected query (best mean sigmoid score over frames, max over classes), its class-0
    logits per frame and query, and the mask logits of the selected query -- what postprocessing.select_trajectory +
    clip_parallel.pack_record produce with nine torch launches per clip."""
    _need_gpu(pred_cls, pred_masks, records)
    lib = _lib.load()
    T, B, Q, K = pred_cls.shape
    if pred_cls.dtype != torch.float32 or pred_cls.stride(3) != 1 and K > 1:
        pred_cls = _f32c(pred_cls)
    pred_masks = _f32c(pred_masks)
    HW = pred_masks.shape[-2] * pred_masks.shape[-1]
    if (records.dtype != torch.float32 or records.dim() != 2 or records.shape[0] != B or records.stride(1) != 1
            or records.shape[1] < 1 + T * Q + T * HW or tuple(pred_masks.shape[:3]) != (T, B, Q)):
        raise _lib.SocHipError("select_pack: records must be float32 [B, >= 1 + T*Q + T*h*w] with contiguous rows")
    with _timed("select_pack", (T * Q * K * B + 2 * T * HW * B) * 4):
        code = lib.soc_select_pack_f32(pred_cls.data_ptr(), pred_cls.stride(0), pred_cls.stride(1), pred_cls.stride(2),
                                       pred_masks.data_ptr(), records.data_ptr(), records.stride(0), T, B, Q, K, HW, _stream())
    _lib.check(code, "soc_select_pack_f32")


def upsample_merge_labels(mask_logits: Tensor, size: Sequence[int], threshold: float = 0.5,
                          background: float = 0.1) -> Tensor:
    """K6, DAVIS form.  [O,T,h,w] logits of O objects -> uint8 labels [T,H0,W0] (0 = background):
    argmax over {background, sigmoid(upsampled logits) zeroed below threshold}."""
    _need_gpu(mask_logits)
    lib = _lib.load()
    x = _f32c(mask_logits)
    O, T, h, w = x.shape
    H0, W0 = int(size[0]), int(size[1])
    out = torch.empty((T, H0, W0), dtype=torch.uint8, device=x.device)
    with _timed("upsample_merge_labels", x.numel() * 4 + out.numel()):
        code = lib.soc_upsample_merge_labels_u8(x.data_ptr(), out.data_ptr(), O, T, h, w, H0, W0, float(threshold),
                                                float(background), _stream())
    _lib.check(code, "soc_upsample_merge_labels_u8")
    return out


def resize_normalize(frames: Tensor, size: Sequence[int], tables_x, tables_y, mean: Sequence[float],
                     std: Sequence[float], return_u8: bool = False):
    """K9.  frames [T,H0,W0,3] uint8 -> [T,3,h,w] float32 = Normalize(ToTensor(PIL bilinear resize)).
    tables_* = (bounds int32 [n,2], coeffs int32 [n,ksize]) device tensors from clip_io.resample_tables."""
    _need_gpu(frames, tables_x[0], tables_x[1], tables_y[0], tables_y[1])
    lib = _lib.load()
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise _lib.SocHipError("resize_normalize: frames must be uint8 [T,H0,W0,3]")
    frames = frames.contiguous()
    T, H0, W0, _ = frames.shape
    h, w = int(size[0]), int(size[1])
    (bx, kx), (by, ky) = tables_x, tables_y
    for tname, t_, n in (("bounds_x", bx, w), ("coeffs_x", kx, w), ("bounds_y", by, h), ("coeffs_y", ky, h)):
        if t_.dtype != torch.int32 or not t_.is_contiguous() or t_.shape[0] != n:
            raise _lib.SocHipError(f"resize_normalize: {tname} must be contiguous int32 with {n} rows")
    out = torch.empty((T, 3, h, w), dtype=torch.float32, device=frames.device)
    out_u8 = torch.empty((T, h, w, 3), dtype=torch.uint8, device=frames.device) if return_u8 else None
    need = lib.soc_resize_workspace_bytes(T, H0, W0, h, w)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=frames.device)
    m = (C.c_float * 3)(*(float(v) for v in mean))
    s = (C.c_float * 3)(*(float(v) for v in std))
    with _timed("resize_normalize", frames.numel() + out.numel() * 4):
        code = lib.soc_resize_normalize_u8_f32(frames.data_ptr(), out.data_ptr(),
                                               None if out_u8 is None else out_u8.data_ptr(), T, H0, W0, h, w,
                                               bx.data_ptr(), kx.data_ptr(), kx.shape[1], by.data_ptr(),
                                               ky.data_ptr(), ky.shape[1], m, s, ws.data_ptr(), need, _stream())
    _lib.check(code, "soc_resize_normalize_u8_f32")
    return (out, out_u8) if return_u8 else out


def groupnorm_tokens_supported(C_: int, groups: int) -> bool:
    cpg, nvec = C_ // max(groups, 1), C_ // 4
    whole = cpg % 4 == 0 and ((cpg // 4) & (cpg // 4 - 1)) == 0
    return (groups > 0 and C_ % groups == 0 and C_ <= 256 and C_ % 4 == 0 and (nvec & (nvec - 1)) == 0
            and (whole or cpg == 2) and groups <= 64)


def groupnorm_tokens(x: Tensor, weight: Tensor, bias: Tensor, groups: int, eps: float = 1e-5,
                     relu: bool = False) -> Tensor:
    """K10.  GroupNorm (+ ReLU) of token-major activations: x [N,S,C] -> [N,S,C], statistics per (n, group) over
    S * C/groups elements (what nn.GroupNorm computes on the '(n) c h w' view of the same data)."""
    _need_gpu(x, weight, bias)
    lib = _lib.load()
    x = _f32c(x)
    N, S, C_ = x.shape
    out = torch.empty_like(x)
    need = lib.soc_groupnorm_tokens_workspace_bytes(N, S, C_, groups)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=x.device)
    with _timed("groupnorm_tokens", 3 * x.numel() * 4):
        code = lib.soc_groupnorm_tokens_f32(x.data_ptr(), _f32c(weight).data_ptr(), _f32c(bias).data_ptr(),
                                            out.data_ptr(), N, S, C_, int(groups), float(eps), int(relu), ws.data_ptr(),
                                            need, _stream())
    _lib.check(code, "soc_groupnorm_tokens_f32")
    return out


def patch_merge_layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """K11.  x [B,D,H,W,C] -> LayerNorm(concat of the 2x2 spatial neighbours) [B,D,ceil(H/2),ceil(W/2),4C]."""
    _need_gpu(x, weight, bias)
    lib = _lib.load()
    x = _f32c(x)
    B, D, H, W, C_ = x.shape
    out = torch.empty((B, D, (H + 1) // 2, (W + 1) // 2, 4 * C_), dtype=torch.float32, device=x.device)
    with _timed("patch_merge_layernorm", x.numel() * 4 + out.numel() * 4):
        code = lib.soc_patch_merge_layernorm_f32(x.data_ptr(), _f32c(weight).data_ptr(), _f32c(bias).data_ptr(),
                                                 out.data_ptr(), B * D, H, W, C_, float(eps), _stream())
    _lib.check(code, "soc_patch_merge_layernorm_f32")
    return out


def patch_embed_supported(frames: Tensor, weight: Tensor) -> bool:
    return (frames.is_cuda and frames.dtype == torch.float32 and frames.dim() == 4 and frames.shape[1] == 3
            and weight.shape[0] in (96, 128) and tuple(weight.shape[1:]) == (3, 1, 4, 4))


def patch_embed_layernorm(frames: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor,
                          eps: float = 1e-5) -> Tensor:
    """K21.  frames [N,3,H,W], conv weight [C,3,1,4,4] -> LayerNorm(conv4x4/4(frames)) [N,ceil(H/4),ceil(W/4),C]."""
    _need_gpu(frames, weight, bias, gamma, beta)
    lib = _lib.load()
    frames = _f32c(frames)
    N, _, H, W = frames.shape
    C_ = weight.shape[0]
    out = torch.empty((N, (H + 3) // 4, (W + 3) // 4, C_), dtype=torch.float32, device=frames.device)
    with _timed("patch_embed_layernorm", frames.numel() * 4 + out.numel() * 4):
        code = lib.soc_patch_embed_layernorm_f32(frames.data_ptr(), _f32c(weight).data_ptr(),
                                                 None if bias is None else _f32c(bias).data_ptr(),
                                                 _f32c(gamma).data_ptr(), _f32c(beta).data_ptr(), out.data_ptr(),
                                                 N, H, W, C_, float(eps), _stream())
    _lib.check(code, "soc_patch_embed_layernorm_f32")
    return out


def linear_act(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: str = "none") -> Tensor:
    """K12.  act(x @ weight.T + bias) with act in {"none", "relu", "gelu"} (gelu = exact erf form) in one tiled
    MFMA GEMM; x [..., K] with K % 16 == 0, weight [N, K] with N % 4 == 0."""
    _need_gpu(x, weight, bias)
    lib = _lib.load()
    x, w = _f32c(x), _f32c(weight)
    K = x.shape[-1]
    M, N = x.numel() // K, w.shape[0]
    out = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
    with _timed("linear_act", (M * K + N * K + M * N) * 4):
        code = lib.soc_linear_act_f32(x.data_ptr(), w.data_ptr(), None if bias is None else _f32c(bias).data_ptr(),
                                      out.data_ptr(), M, N, K, {"none": 0, "relu": 1, "gelu": 2}[act], _stream())
    _lib.check(code, "soc_linear_act_f32")
    return out


def linear_act_multi(x: Tensor, layers: Sequence[Tuple[Tensor, Optional[Tensor]]], add: Optional[Tensor] = None,
                     act: str = "none") -> List[Tensor]:
    """K12, several outputs.  [act((x + add) @ w.T + b) for (w, b) in layers] (at most four layers) in one tiled
    MFMA GEMM launch; `add` (optional) has x's shape."""
    _need_gpu(x, add, *(w for w, _ in layers))
    lib = _lib.load()
    x = _f32c(x)
    K = x.shape[-1]
    M = x.numel() // K
    add_ptr = None
    if add is not None:
        add = _f32c(add.expand_as(x))
        add_ptr = add.data_ptr()
    n = len(layers)
    ws = [_f32c(w) for w, _ in layers]
    bs = [None if b is None else _f32c(b) for _, b in layers]
    outs = [torch.empty(*x.shape[:-1], w.shape[0], dtype=torch.float32, device=x.device) for w in ws]
    vp = C.c_void_p * n
    with _timed("linear_act", (M * K * (1 + (add is not None)) + sum(w.numel() + M * w.shape[0] for w in ws)) * 4):
        code = lib.soc_linear_act_multi_f32(x.data_ptr(), add_ptr, n, vp(*(w.data_ptr() for w in ws)),
                                            vp(*(None if b is None else b.data_ptr() for b in bs)),
                                            vp(*(o.data_ptr() for o in outs)), (C.c_int * n)(*(w.shape[0] for w in ws)),
                                            M, K, {"none": 0, "relu": 1, "gelu": 2}[act], _stream())
    _lib.check(code, "soc_linear_act_multi_f32")
    return outs


WS_LINEAR_K = (96, 128, 192, 256, 384, 512)
WS_SPLIT_LN_K = (96, 128, 192)      # widths K13b (bf16 matrix cores) covers with a LayerNorm in front; 384 / 512 without


def ws_linear_supported(x: Tensor, weight: Tensor, has_ln: bool) -> bool:
    """True when K13 takes linear(x, weight): CUDA fp32, K one of its widths (LayerNorm: K <= 256), N % 16 == 0."""
    N, K = weight.shape
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and K in WS_LINEAR_K
            and N % 16 == 0 and (not has_ln or K <= 256) and x.shape[-1] == K)


def ws_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, ln: Optional[Tuple[Tensor, Tensor, float]] = None,
              residual: Optional[Tensor] = None, act: str = "none") -> Tensor:
    """K13: act(LN(x) @ weight.T + bias) + residual, every part optional; ln = (gamma, beta, eps).
    x [..., K] -> [..., N]; residual has the output's shape."""
    _need_gpu(x, weight, bias, residual, *(ln[:2] if ln else ()))
    lib = _lib.load()
    x, weight = _f32c(x), _f32c(weight)
    N, K = weight.shape
    M = x.numel() // K
    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _f32c(residual)
        if residual.shape != out.shape:
            raise _lib.SocHipError(f"ws_linear: residual shape {tuple(residual.shape)} != output {tuple(out.shape)}")
    g = be = None
    eps = 0.0
    if ln is not None:
        g, be, eps = _f32c(ln[0]), _f32c(ln[1]), float(ln[2])
    b = _f32c(bias) if bias is not None else None
    code = {"none": 0, "relu": 1, "gelu": 2}[act]
    work = 2.0 * M * N * K
    _rec = op_profile.recording("k13")
    if _rec is not None:
        _rec.append(dict(x=x, weight=weight, bias=bias, ln=ln, residual=residual, act=act))
    with _timed("ws_linear", work):
        rc = lib.soc_ws_linear_f32(x.data_ptr(), g.data_ptr() if g is not None else None,
                                   be.data_ptr() if be is not None else None, eps, weight.data_ptr(),
                                   b.data_ptr() if b is not None else None,
                                   residual.data_ptr() if residual is not None else None, out.data_ptr(), M, N, K, code,
                                   int(k13_split_enabled()), _stream())      # split: K13b where it covers the width
    _lib.check(rc, "soc_ws_linear_f32")
    return out


# ---------------------------------------------------------------------------------------------
# K20: f32 linear layers on the bf16 matrix cores by exact operand splitting
import os as _os

# The arithmetic mode of a forward and the routing rules that hang on it live in matmul_mode.py, the cache of packed weight images
# in derived_cache.py; both are re-exported here (callers and tests use hot_ops.<name>).
from .derived_cache import DerivedCache  # noqa: E402,F401
from .matmul_mode import (DEFAULT_MATMUL_MODE, _SPLIT_OFF, _mode_tls, k1_split_enabled, k13_split_enabled,  # noqa: E402,F401
                          matmul_mode, split_enabled, split_wins, use_matmul_mode)


_SPLIT_ACT = {"none": 0, "relu": 1, "gelu": 2}
SPLIT_TILES = {0: (128, 256), 1: (256, 128), 2: (128, 128), 3: (256, 96), 4: (128, 64)}
_split_cache = DerivedCache()


def split_pack(weight: Tensor, bias: Optional[Tensor] = None, gamma: Optional[Tensor] = None,
               beta: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """(K20 weight image, effective bias, column sums) of a layer, built once and re-built when a tensor involved is
    modified in place (load_state_dict) or replaced.  With (gamma, beta) the LayerNorm in front of the layer is folded in:
        LN(x) w^T + b = rstd (x (w diag(gamma))^T - mean colsum) + (b + w beta)
    the image holds w * gamma (one f32 rounding per weight), the bias is b + w beta and colsum[n] = sum_k (w * gamma)[n, k]
    of exactly the f32 values in the image (both summed in f64, rounded once)."""
    _need_gpu(weight, bias, gamma, beta)
    lib = _lib.load()

    def build():
        w = _f32c(weight.detach())
        N, K = w.shape
        eff_bias = _f32c(bias.detach()) if bias is not None else None
        colsum = None
        if gamma is not None:
            eb = w.double() @ beta.detach().double()
            if bias is not None:
                eb = eb + bias.detach().double()
            eff_bias = eb.float().contiguous()
            w = (w * gamma.detach()[None, :]).contiguous()
            colsum = w.double().sum(1).float().contiguous()
        packed = torch.empty(lib.soc_linear_split_packed_bytes(N, K), dtype=torch.uint8, device=w.device)
        _lib.check(lib.soc_linear_split_pack_f32(w.data_ptr(), packed.data_ptr(), N, K, _stream()), "soc_linear_split_pack_f32")
        return packed, eff_bias, colsum
    return _split_cache.get((weight, bias, gamma, beta), build, extra=(bias is None, gamma is None))


def row_stats(x: Tensor, eps: float) -> Tensor:
    """[..., K] -> [rows, 2] (mean, rstd) as nn.LayerNorm computes them (K % 4 == 0, K <= 1024)."""
    _need_gpu(x)
    lib = _lib.load()
    x = _f32c(x)
    K = x.shape[-1]
    M = x.numel() // K
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    with _timed("row_stats", 4.0 * M * K):
        rc = lib.soc_row_stats_f32(x.data_ptr(), stats.data_ptr(), M, K, float(eps), _stream())
    _lib.check(rc, "soc_row_stats_f32")
    return stats


def split_tile_for(M: int, N: int, K: int) -> int:
    """Tile configuration of K20 for an [M, K] x [N, K]^T layer: the widest tile that wastes no columns, unless that
    leaves the 256 CUs with less than about two tiles each (then the 128 x 128 / 128 x 64 tiles)."""
    def tiles(cfg):
        bm, bn = SPLIT_TILES[cfg]
        return -(-M // bm) * -(-N // bn)
    if N % 96 == 0 and N % 128 != 0:
        cfg = 3
    elif N % 256 == 0:
        cfg = 0
    elif N % 128 == 0:
        cfg = 1
    elif N <= 64:
        cfg = 4
    else:
        cfg = 0
    if tiles(cfg) < 384 and N % 64 == 0:      # also N = 192: 226 tiles of 256 x 96 leave CUs idle (41.9 -> 30.9 us at stage 1)
        cfg = 2
    if tiles(cfg) < 256 and cfg == 2:
        cfg = 4
    return cfg


def small_attention_supported(q: Tensor, n_heads: int, mask: Optional[Tensor]) -> bool:
    """K25 takes softmax(q k^T * scale + mask) v of a [B, L, E] self-attention: CUDA fp32, L <= 64, head dim 32 / 64, mask None or
    additive float [B, 1, 1 | L, L]."""
    if not (q.is_cuda and q.dtype == torch.float32 and q.dim() == 3 and q.shape[-1] % n_heads == 0):
        return False
    B, L, E = q.shape
    if L > 64 or E // n_heads not in (32, 64):
        return False
    return mask is None or (mask.dtype == torch.float32 and mask.dim() == 4 and mask.shape[0] in (1, B) and mask.shape[1] == 1
                            and mask.shape[2] in (1, L) and mask.shape[3] == L)


def small_attention(q: Tensor, k: Tensor, v: Tensor, n_heads: int, mask: Optional[Tensor] = None,
                    scale: Optional[float] = None) -> Tensor:
    """K25: multi-head self-attention core on token-major [B, L, E] tensors (heads = E / n_heads wide column blocks), additive
    float mask [B | 1, 1, L | 1, L] or None -> [B, L, E].  Same result as F.scaled_dot_product_attention on the
    [B, H, L, D] views."""
    _need_gpu(q, k, v, mask)
    lib = _lib.load()
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    B, L, E = q.shape
    D = E // n_heads
    out = torch.empty_like(q)
    mb = mq = 0
    mp = None
    if mask is not None:
        mask = _f32c(mask)
        mp = mask.data_ptr()
        mb = mask.stride(0) if mask.shape[0] > 1 else 0
        mq = mask.stride(2) if mask.shape[2] > 1 else 0
    with _timed("small_attn", 4.0 * B * n_heads * L * L * D):
        rc = lib.soc_small_attn_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(), B, L, n_heads, D,
                                    float(scale) if scale is not None else D ** -0.5, mb, mq, _stream())
    _lib.check(rc, "soc_small_attn_f32")
    return out


XS_LINEAR_K = (192, 256, 384, 512, 768, 1024)      # input widths K24 is built for
_XS_NCT = (18, 16, 12, 8, 6, 4)         # column tiles per range it is built for
_xs_cache = DerivedCache()
def xs_linear_supported(x, weight) -> bool:
    """K24 takes act(LN(x) weight^T + bias) + residual: CUDA fp32, input width 192 / 256 / 384 / 512 / 768 / 1024, an output width whose 16-
    column tiles divide into ranges of a built size, split arithmetic on (SOC_SPLIT_OFF=k24 switches it off)."""
    N, K = weight.shape
    ctp = 2 if K <= 256 else 1
    top = 8 if K > 768 else 18              # K = 1024 keeps 384 registers of x fragments: ranges of at most 8 column tiles
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and split_enabled()
            and "k24" not in _SPLIT_OFF and x.shape[-1] == K and K in XS_LINEAR_K and N % 32 == 0
            and any((N // 16) % n == 0 and n % ctp == 0 and n <= top for n in _XS_NCT))


def _xs_packed(weight: Tensor) -> Tensor:
    lib = _lib.load()
    N, K = weight.shape

    def build():
        packed = torch.empty(lib.soc_xs_linear_packed_bytes(N, K), dtype=torch.uint8, device=weight.device)
        wc = _f32c(weight.detach())
        _lib.check(lib.soc_xs_linear_pack_f32(wc.data_ptr(), packed.data_ptr(), N, K, _stream()), "soc_xs_linear_pack_f32")
        return packed
    return _xs_cache.get((weight,), build)


def xs_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, ln: Optional[Tuple[Tensor, Tensor, float]] = None,
              residual: Optional[Tensor] = None, act: str = "none", cut: Optional[Tuple[int, int]] = None) -> Tensor:
    """K24: act(LN(x) @ weight.T + bias) + residual, every part optional; ln = (gamma, beta, eps); x [..., K] -> [..., N].
    `cut` = (workgroup rows, column ranges) forces the decomposition (tests, probes)."""
    _need_gpu(x, weight, bias, residual, *(ln[:2] if ln else ()))
    lib = _lib.load()
    x = _f32c(x)
    N, K = weight.shape
    M = x.numel() // K
    packed = _xs_packed(weight)
    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _f32c(residual)
        if residual.shape != out.shape:
            raise _lib.SocHipError(f"xs_linear: residual shape {tuple(residual.shape)} != output {tuple(out.shape)}")
    g = be = None
    eps = 0.0
    if ln is not None:
        g, be, eps = _f32c(ln[0]), _f32c(ln[1]), float(ln[2])
    b = _f32c(bias) if bias is not None else None
    _rec = op_profile.recording("k24")
    if _rec is not None:
        _rec.append(dict(x=x, weight=weight, bias=bias, ln=ln, residual=residual, act=act, cut=cut))
    ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    nrg, ncr = cut if cut is not None else (0, 0)
    with _timed("xs_linear", 2.0 * M * N * K):
        rc = lib.soc_xs_linear_f32(x.data_ptr(), packed.data_ptr(), ptr(b), ptr(g), ptr(be), eps, ptr(residual), out.data_ptr(),
                                   M, N, K, {"none": 0, "relu": 1, "gelu": 2}[act], int(nrg), int(ncr), _stream())
    _lib.check(rc, "soc_xs_linear_f32")
    return out


def xs_linear_plan(M: int, N: int, K: int) -> Tuple[int, int, int]:
    """(workgroup rows, column ranges, column tiles per range) K24 cuts an [M, K] x [N, K]^T layer into."""
    lib = _lib.load()
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.check(lib.soc_xs_linear_plan(M, N, K, C.byref(a), C.byref(b), C.byref(c), _plan_stream()), "soc_xs_linear_plan")
    return a.value, b.value, c.value


# Output placement: graph_runner.PipelinedClipGraph hands the head two of its static buffers (the stage-0 token map and the
# encoder memory, 44 + 40 MB) so that the kernels producing them write there directly instead of being copied there afterwards.
_placed: dict = {}


def place_output(tag: str, tensor: Optional[Tensor]) -> None:
    """Ask the producer of `tag` ("swin0": last block of Video-Swin stage 0; "encoder_memory": last encoder layer) to write its
    result into `tensor` (contiguous fp32, the result's shape); None withdraws the request."""
    if tensor is None:
        _placed.pop(tag, None)
    else:
        _placed[tag] = tensor


def placed(tag: Optional[str], like: Tensor) -> Optional[Tensor]:
    t = _placed.get(tag) if tag is not None else None
    if t is None or t.shape != like.shape or t.dtype != like.dtype or t.device != like.device or not t.is_contiguous():
        return None
    return t


MLP_SPLIT_C = (96, 128, 192, 256, 384, 512)  # model widths K23 is built for
_MLP_ACT = {"relu": 1, "gelu": 2}
_mlp_cache = DerivedCache()




# soc_mlp_split_max_hidden(C) of csrc/mlp_split.hip as a table: routing predicates read shapes only and must not need the
# native library (routes.table() runs on a CPU-only checkout).  tests/test_c_abi.py and the GPU suite check the
# table against the library whenever it is there.
MLP_SPLIT_MAX_HIDDEN = {96: 16192, 128: 16128, 192: 9856, 256: 3584, 384: 3328, 512: 3072}


def mlp_split_max_hidden(Cw: int) -> int:
    """Largest hidden width K23 holds at model width Cw (its b1 range shares the 160 KB of LDS with the weight ring): beyond it
    the launch would return SOC_EUNSUPPORTED, so the router must not send the layer there.  Shape-only: a table."""
    return MLP_SPLIT_MAX_HIDDEN.get(int(Cw), 0)


def mlp_split_supported(x: Tensor, w1: Tensor, w2: Tensor) -> bool:
    """K23 takes act(LN(x) w1^T + b1) w2^T + b2 (+ residual): CUDA fp32, model width 96 / 128 / 192 / 256 / 384 / 512, hidden width a
    multiple of 32 up to mlp_split_max_hidden(width), split arithmetic on (SOC_SPLIT_OFF=mlp switches it off)."""
    Cw = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and split_enabled() and "mlp" not in _SPLIT_OFF and Cw in MLP_SPLIT_C
            and tuple(w1.shape[1:]) == (Cw,) and tuple(w2.shape) == (Cw, w1.shape[0]) and w1.shape[0] % 32 == 0
            and w1.shape[0] <= mlp_split_max_hidden(Cw)
            and w1.dtype == torch.float32 and w2.dtype == torch.float32)


def _mlp_packed(w1: Tensor, w2: Tensor) -> Tensor:
    """The K23 weight image of (w1, w2): built once, rebuilt after an in-place update, freed with the model."""
    lib = _lib.load()
    F_, C_ = w1.shape

    def build():
        packed = torch.empty(lib.soc_mlp_split_packed_bytes(C_, F_), dtype=torch.uint8, device=w1.device)
        w1c, w2c = _f32c(w1.detach()), _f32c(w2.detach())
        _lib.check(lib.soc_mlp_split_pack_f32(w1c.data_ptr(), w2c.data_ptr(), packed.data_ptr(), C_, F_, _stream()),
                   "soc_mlp_split_pack_f32")
        return packed
    return _mlp_cache.get((w1, w2), build)


def mlp_split(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, act: str = "gelu",
              ln: Optional[Tuple[Tensor, Tensor, float]] = None, residual: Optional[Tensor] = None,
              out: Optional[Tensor] = None, cut: Optional[Tuple[int, int]] = None, variant: int = 0,
              post_ln: Optional[Tuple[Tensor, Tensor, float]] = None, return_sum: bool = False, residual_ln: bool = False):
    """K23: LN2(act(LN(x) @ w1.T + b1) @ w2.T + b2 + residual) in one launch, the hidden layer in registers; ln / post_ln =
    (gamma, beta, eps) or None, act "relu" | "gelu".  With post_ln and return_sum the result is (sum in front of LN2, LN2(sum)).
    residual_ln: the shortcut is LN(x) -- the LayerNorm `ln` -- instead of x itself (the encoder's norm1, whose result is both the
    block's input and its shortcut); `residual` must then be x itself (same storage), anything else is refused.  `cut` = (workgroup rows, hidden ranges) forces one launch with that decomposition
    (tests, probes); by default the library plans whole rounds + a split tail."""
    _need_gpu(x, w1, b1, w2, b2, residual, *(ln[:2] if ln else ()), *(post_ln[:2] if post_ln else ()))
    lib = _lib.load()
    x = _f32c(x)
    F_, C_ = w1.shape
    M = x.numel() // C_
    packed = _mlp_packed(w1, w2)
    g2 = be2 = None
    eps2 = 0.0
    if post_ln is not None:
        g2, be2, eps2 = _f32c(post_ln[0]), _f32c(post_ln[1]), float(post_ln[2])
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise _lib.SocHipError("mlp_split: `out` must be a contiguous float32 tensor of the input's shape on its device")
    if residual is not None:
        residual = _f32c(residual)
        if residual.shape != x.shape:
            raise _lib.SocHipError("mlp_split: residual shape differs from the input's")
    g = be = None
    eps = 0.0
    if ln is not None:
        g, be, eps = _f32c(ln[0]), _f32c(ln[1]), float(ln[2])
    ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    b1c, b2c = _f32c(b1), _f32c(b2)
    if return_sum and post_ln is None:
        raise _lib.SocHipError("mlp_split: return_sum needs post_ln (without it the result is the sum)")
    if residual_ln and (ln is None or residual is None):
        raise _lib.SocHipError("mlp_split: residual_ln needs both ln and residual")
    if residual_ln and residual.data_ptr() != x.data_ptr():
        raise _lib.SocHipError("mlp_split: residual_ln means 'the shortcut is LN(x)': residual must be x itself")
    osum = torch.empty_like(x) if return_sum else None
    _rec = op_profile.recording("k23")
    if _rec is not None:
        _rec.append(dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, act=act, ln=ln, residual=residual, post_ln=post_ln, cut=cut,
                               return_sum=return_sum, residual_ln=residual_ln))
    if cut is None:
        nbytes = lib.soc_mlp_split_workspace_bytes(M, C_, F_, _stream())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        with _timed("mlp_split", 4.0 * M * F_ * C_):
            rc = lib.soc_mlp_split_f32(x.data_ptr(), packed.data_ptr(), b1c.data_ptr(), b2c.data_ptr(), ptr(g), ptr(be), eps,
                                       ptr(residual), ptr(g2), ptr(be2), eps2, out.data_ptr(), ptr(osum), ptr(ws), nbytes, M, C_,
                                       F_, _MLP_ACT[act], int(residual_ln), _stream())
        _lib.check(rc, "soc_mlp_split_f32")
        return (osum, out) if return_sum else out
    nrg, nfs = cut
    ws = torch.empty(nfs * M * C_, dtype=torch.float32, device=x.device) if nfs > 1 else None
    with _timed("mlp_split", 4.0 * M * F_ * C_):
        rc = lib.soc_mlp_split_variant_f32(x.data_ptr(), packed.data_ptr(), b1c.data_ptr(), b2c.data_ptr(), ptr(g), ptr(be), eps,
                                           ptr(residual), ptr(g2), ptr(be2), eps2, out.data_ptr(), ptr(osum), ptr(ws), M, C_, F_,
                                           _MLP_ACT[act], int(residual_ln), int(nrg), int(nfs), int(variant), _stream())
    _lib.check(rc, "soc_mlp_split_variant_f32")
    return (osum, out) if return_sum else out


def mlp_split_plan(M: int, C_: int, F_: int) -> Tuple[int, int]:
    """(workgroup rows, hidden ranges) K23 would cut M rows into as ONE launch."""
    lib = _lib.load()
    nrg, nfs = C.c_int(0), C.c_int(0)
    _lib.check(lib.soc_mlp_split_plan(M, C_, F_, C.byref(nrg), C.byref(nfs), _plan_stream()), "soc_mlp_split_plan")
    return nrg.value, nfs.value


def linear_split_supported(x: Tensor, weight: Tensor, ln: bool = False) -> bool:
    N, K = weight.shape
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.shape[-1] == K
            and K % 8 == 0 and N % 4 == 0 and (not ln or K <= 1024))


def linear_split(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
                 ln: Optional[Tuple[Tensor, Tensor, float]] = None, residual: Optional[Tensor] = None,
                 act: str = "none", add: Optional[Tensor] = None, mul: Optional[Tensor] = None,
                 tile: Optional[int] = None, stats: Optional[Tensor] = None, split_at: Optional[int] = None):
    """K20: mul * act(LN(x + add) @ weight.T + bias) + residual, every part optional; ln = (gamma, beta, eps).
    f32 in / f32 out with f32-level error (three-way bf16 split of both operands, six exact products, f32 accumulation);
    the LayerNorm is folded into the cached weight image / bias / column sums (split_pack) and applied in the epilogue.
    `stats` may carry precomputed row statistics of x (row_stats).  ln and add cannot be combined (no caller does)."""
    _need_gpu(x, weight, bias, residual, add, mul, *(ln[:2] if ln else ()))
    lib = _lib.load()
    x = _f32c(x)
    N, K = weight.shape
    M = x.numel() // K
    bias_in = bias
    packed, bias, colsum = split_pack(weight, bias, *(ln[:2] if ln is not None else (None, None)))
    out2 = None
    if split_at is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    else:       # two stacked layers on the same input -> two contiguous outputs
        out = torch.empty(x.shape[:-1] + (split_at,), dtype=torch.float32, device=x.device)
        out2 = torch.empty(x.shape[:-1] + (N - split_at,), dtype=torch.float32, device=x.device)
        if residual is not None or mul is not None:
            raise _lib.SocHipError("linear_split: split_at cannot be combined with residual / mul")
    for name, t in (("residual", residual), ("mul", mul)):
        if t is not None and t.shape != out.shape:
            raise _lib.SocHipError(f"linear_split: {name} shape {tuple(t.shape)} != output {tuple(out.shape)}")
    if add is not None and add.shape != x.shape:
        raise _lib.SocHipError(f"linear_split: add shape {tuple(add.shape)} != input {tuple(x.shape)}")
    if ln is not None:
        if add is not None:
            raise _lib.SocHipError("linear_split: LayerNorm in front of the layer cannot be combined with `add`")
        if stats is None:
            stats = row_stats(x, float(ln[2]))
    residual = _f32c(residual) if residual is not None else None
    mul = _f32c(mul) if mul is not None else None
    add = _f32c(add) if add is not None else None
    b = _f32c(bias) if bias is not None else None
    if M == 0:
        return out if out2 is None else (out, out2)
    cfg = split_tile_for(M, N, K) if tile is None else int(tile)
    _rec = op_profile.recording("k20")
    if _rec is not None:
        _rec.append(dict(x=x, weight=weight, bias=bias_in, ln=ln, residual=residual, act=act, add=add, mul=mul,
                               tile=cfg, stats=stats, split_at=split_at))
    ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    with _timed("linear_split", 2.0 * M * N * K):
        rc = lib.soc_linear_split_f32(x.data_ptr(), ptr(add), ptr(stats) if ln is not None else None, ptr(colsum),
                                      packed.data_ptr(), ptr(b), ptr(residual), ptr(mul), out.data_ptr(), ptr(out2),
                                      int(split_at or 0), M, N, K, _SPLIT_ACT[act], cfg, _stream())
    _lib.check(rc, "soc_linear_split_f32")
    return out if out2 is None else (out, out2)
