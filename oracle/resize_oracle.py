"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's frame pre-processing:
PIL bilinear resize -> ToTensor -> Normalize (infer_refytb.py:33-38,193-201; datasets/transforms.py:186-216).

The resize itself lives in a third-party dependency of the reference, Pillow (`Image.resize(size, BILINEAR)`,
libImaging/Resample.c, any version >= 4; 12.2.0 is installed in this image and is what the oracle is pinned
against in tests/test_oracle_vs_golden.py): two separable passes, horizontal first, anti-aliasing
triangle filter whose support grows with the reduction factor, coefficients in 8.22 fixed point,
round-half-up and clip to uint8 after EACH pass.  Vectorised numpy; independent of the product's
clip_io.resample_tables (written from the same published algorithm, different code).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def size_with_aspect_ratio(w: int, h: int, size: int, max_size=None):
    """datasets/transforms.py:189-207 -> (oh, ow)"""
    if max_size is not None:
        mn, mx = float(min((w, h))), float(max((w, h)))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        return (int(size * h / w), size)
    return (size, int(size * w / h))


def coeffs_1d(in_size: int, out_size: int):
    """Resample.c precompute_coeffs (bilinear) + normalize_coeffs_8bpc, vectorised over output index."""
    scale = in_size / out_size
    fscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((centers - support + 0.5).astype(np.int64), 0)            # C cast: truncation
    xmax = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((taps + xmin[:, None] - centers[:, None] + 0.5) * (1.0 / fscale))
    wgt = np.where(arg < 1.0, 1.0 - arg, 0.0)
    wgt = np.where(taps < n[:, None], wgt, 0.0)
    tot = np.zeros(out_size)
    for x in range(ksize):                      # Pillow accumulates ww left to right
        tot = tot + wgt[:, x]
    wgt = np.where(tot[:, None] != 0.0, wgt / np.where(tot == 0.0, 1.0, tot)[:, None], wgt)
    fixed = np.where(wgt < 0, (-0.5 + wgt * (1 << PRECISION_BITS)).astype(np.int64),
                     (0.5 + wgt * (1 << PRECISION_BITS)).astype(np.int64))
    return xmin, n, fixed, ksize


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """one resampling pass of uint8 [H,W,C] along `axis` (0 = vertical, 1 = horizontal)"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)         # [in, other, C]
    xmin, n, k, ksize = coeffs_1d(src.shape[0], out_size)
    acc = np.full((out_size, *src.shape[1:]), 1 << (PRECISION_BITS - 1), dtype=np.int64)
    for x in range(ksize):
        idx = np.minimum(xmin + x, src.shape[0] - 1)        # taps >= n carry a zero coefficient
        acc += src[idx] * k[:, x].reshape(-1, *([1] * (src.ndim - 1)))
    out = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [oh,ow,3] as PIL.Image.resize((ow, oh), BILINEAR).  Pillow skips a pass
    whose size does not change."""
    out = img
    if ow != img.shape[1]:
        out = _pass(out, ow, 1)
    if oh != img.shape[0]:
        out = _pass(out, oh, 0)
    return out


def preprocess_clip(frames: np.ndarray, size: int = 360, max_size=640, mean=(0.485, 0.456, 0.406),
                    std=(0.229, 0.224, 0.225)):
    """uint8 [T,H0,W0,3] -> (float32 [T,3,h,w], uint8 [T,h,w,3]): resize, x/255, (x-mean)/std in fp32
    with torchvision's operation order (to_tensor: .div(255); normalize: .sub_(mean).div_(std))."""
    T, H0, W0, _ = frames.shape
    oh, ow = size_with_aspect_ratio(W0, H0, size, max_size)
    small = np.stack([resize_bilinear_u8(f, oh, ow) for f in frames])
    x = small.astype(np.float32) / np.float32(255)
    x = (x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)), small
