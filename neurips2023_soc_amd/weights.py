"""Deterministic synthetic checkpoint for SOC (SURVEY.md section 8d "Weights").

There is no network on either box and a 700 MB checkpoint cannot travel, so every
tensor of the ``state_dict`` is regenerated from ``(seed, key, shape)`` by an integer-only
generator: FNV-1a over the key name -> splitmix64 counter stream -> four 16-bit lanes summed
(Irwin-Hall, n=4) and scaled to unit variance.  No libm call is involved, so the reference
model in the build container and the HIP model on the GPU box see bit-identical weights.

The recipe deliberately differs from the reference's default init where that init is
degenerate for testing (reference: models/ops/modules/ms_deform_attn.py:63-77 zeroes the
sampling-offset / attention-weight matrices, models/soc.py:83-84 zeroes the last box layer),
see SURVEY.md Appendix B.2.  ``controller.layers.2`` is scaled down so mask logits have a
trained-like O(10) magnitude (reference feeds +-640 px relative coordinates into the
dynamic convolution, models/soc.py:431-440).
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)

# scale applied to controller.layers.2.{weight,bias}; tuned so that max|mask logit| ~ 10-30
CONTROLLER_OUT_SCALE = 0.3
# added to the last dynamic-conv bias so that thresholded masks are not almost all-background
CONTROLLER_LOGIT_SHIFT = 5.0


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + _GOLDEN) & _MASK64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        return z ^ (z >> np.uint64(31))


def unit_normal(seed: int, key: str, n: int) -> np.ndarray:
    """n float32 values, zero mean / unit variance, platform independent."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    # var(sum of 4 U{0..65535}) = 4 * (65536^2 - 1) / 12
    inv_sigma = 1.0 / math.sqrt((65536.0 ** 2 - 1.0) / 3.0)
    m16 = np.uint64(0xFFFF)
    out = np.empty(n, dtype=np.float32)
    chunk = 1 << 18  # stay cache resident; the result is independent of the chunking
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        with np.errstate(over="ignore"):
            ctr = (np.arange(lo, hi, dtype=np.uint64) * _GOLDEN + base) & _MASK64
        bits = _splitmix64(ctr)
        s = ((bits & m16) + ((bits >> np.uint64(16)) & m16)
             + ((bits >> np.uint64(32)) & m16) + (bits >> np.uint64(48))).astype(np.int64)
        s -= 2 * 65535
        out[lo:hi] = s.astype(np.float64) * inv_sigma
    return out


_NORM_PAT = re.compile(
    r"(^|\.)(norm\d*|gn\d+|layer_norm|LayerNorm|decoder_norm|norm)\.(weight|bias)$")


def _is_norm_key(key: str, shape: Tuple[int, ...]) -> bool:
    if len(shape) != 1:
        return False
    if _NORM_PAT.search(key):
        return True
    # input_proj.N.1 is the GroupNorm of nn.Sequential(conv, gn)  (reference models/soc.py:61-69)
    return re.search(r"input_proj\.\d+\.1\.(weight|bias)$", key) is not None


def _msda_offset_bias(n_heads: int, n_levels: int, n_points: int) -> np.ndarray:
    """Deterministic star-shaped offset bias: heads point in 8 directions, point p at radius p+1.

    Follows the geometric intent of reference models/ops/modules/ms_deform_attn.py:63-71 with
    exactly representable directions (max-norm unit vectors) so no trigonometry is needed.
    """
    assert n_heads == 8, "synthetic offset bias is defined for 8 heads"
    dirs = np.array([[1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, -1], [1, -1]],
                    dtype=np.float32)
    out = np.zeros((n_heads, n_levels, n_points, 2), dtype=np.float32)
    for p in range(n_points):
        out[:, :, p, :] = dirs[:, None, :] * float(p + 1)
    return out.reshape(-1)


def make_tensor(seed: int, key: str, shape: Tuple[int, ...]) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    # transformer.decoder.bbox_embed is the SAME module object as bbox_embed (reference
    # models/soc.py:91-95), so a real checkpoint stores identical tensors under both names.
    if key.startswith("transformer.decoder.bbox_embed."):
        key = key[len("transformer.decoder."):]
    z = unit_normal(seed, key, n)

    if _is_norm_key(key, shape):
        arr = (1.0 + 0.02 * z) if key.endswith("weight") else 0.02 * z
    elif key.endswith("relative_position_bias_table"):
        arr = 0.2 * z
    elif key.endswith("sampling_offsets.bias"):
        arr = _msda_offset_bias(8, shape[0] // (8 * 4 * 2), 4) + 0.05 * z
    elif key.endswith("sampling_offsets.weight"):
        arr = 0.05 * z
    elif key.endswith("attention_weights.weight"):
        arr = 0.05 * z
    elif key.endswith("level_embed") or key.endswith("query_embed.weight") \
            or key.endswith("fq_pos.weight"):
        arr = z
    elif "embeddings" in key and key.endswith("weight") and len(shape) == 2:
        arr = 0.5 * z  # RoBERTa embedding tables (followed by LayerNorm)
    elif re.search(r"bbox_embed\.\d+\.layers\.2\.weight$", key):
        arr = 0.02 * z
    elif re.search(r"bbox_embed\.\d+\.layers\.2\.bias$", key):
        arr = 0.02 * z
        arr = arr.copy()
        arr[2:] -= 2.0  # reference models/soc.py:93 biases w,h logits to -2
    elif key.startswith("class_embed") and key.endswith("bias"):
        arr = -math.log(99.0) + 0.5 * z  # prior 0.01, reference models/soc.py:80-82
    elif key.endswith("bias") or len(shape) <= 1:
        arr = 0.02 * z
    else:
        fan_in = int(np.prod(shape[1:]))
        arr = z * (1.0 / math.sqrt(fan_in))

    if key.startswith("controller.layers.2."):
        arr = arr * CONTROLLER_OUT_SCALE
        if key.endswith("bias"):
            arr = arr.copy()
            arr[-1] += CONTROLLER_LOGIT_SHIFT  # b2 of the dynamic head: balance the mask signs
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).reshape(shape).copy())


def synthetic_state_dict(shapes: Mapping[str, Iterable[int]], seed: int = 2023,
                         skip_suffixes: Tuple[str, ...] = (
                             "relative_position_index", "position_ids", "token_type_ids"),
                         ) -> Dict[str, torch.Tensor]:
    """Generate every floating-point entry named in ``shapes`` (key -> shape).

    Integer buffers (``relative_position_index`` and the HF position/token-type id buffers)
    are derived by the modules themselves and are skipped; load with ``strict=False``.
    """
    out: Dict[str, torch.Tensor] = {}
    for key in sorted(shapes):
        if key.endswith(skip_suffixes):
            continue
        out[key] = make_tensor(seed, key, tuple(shapes[key]))
    return out


def float_shapes_of(module: torch.nn.Module) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in module.state_dict().items() if v.is_floating_point()}


def load_synthetic(module: torch.nn.Module, seed: int = 2023) -> Dict[str, torch.Tensor]:
    """Fill ``module`` in place with the synthetic checkpoint; returns the generated dict."""
    sd = synthetic_state_dict(float_shapes_of(module), seed)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    bad = [k for k in missing if not k.endswith(
        ("relative_position_index", "position_ids", "token_type_ids"))]
    if bad or unexpected:
        raise RuntimeError(f"synthetic checkpoint mismatch: missing={bad} unexpected={unexpected}")
    return sd


def synthetic_clip(seed: int, T: int = 8, H: int = 360, W: int = 640) -> torch.Tensor:
    """Normalised-image-space clip [T,3,H,W] (SURVEY.md 8d 'Synthetic inputs')."""
    z = unit_normal(seed, "clip", T * 3 * H * W)
    return torch.from_numpy(z.reshape(T, 3, H, W).copy())


def synthetic_token_ids(seed: int, L: int = 10, vocab: int = 50265) -> torch.Tensor:
    """[1,L] int64 ids: <s>=0 ... </s>=2, the rest uniform in [3, vocab)."""
    base = np.uint64((_fnv1a64("tokens") ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        bits = _splitmix64((np.arange(L, dtype=np.uint64) * _GOLDEN + base) & _MASK64)
    ids = (bits % np.uint64(vocab - 3)).astype(np.int64) + 3
    ids[0] = 0
    ids[-1] = 2
    return torch.from_numpy(ids.reshape(1, L).copy())
