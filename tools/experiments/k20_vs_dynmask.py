#!/usr/bin/env python
"""K4 (dynamic mask head: v_pk_fma_f32 chains with SGPR weights, no LDS) launched on a side stream while the main
stream runs an "aggressor" kernel; K4's output is compared bit for bit with its first value and the deviating elements
are histogrammed by lane (pixel pair index & 63) and by half (even / odd x).

    python tools/experiments/k20_vs_dynmask.py [rounds] [aggressor ...]
aggressors: k20 (split-bf16 linear + GELU, 7360x1536x384), k20_s0 (117760x384x96), k1 (window attention, split-bf16),
            k1f32 (window attention, f32 MFMA), k13 (f32-MFMA weight-stationary linear), lib (rocBLAS/hipBLASLt f32 GEMM),
            none
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neurips2023_soc_amd import hot_ops, _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
modes = sys.argv[2:] or ["none", "k20", "k20_s0", "k1", "k1f32", "k13", "lib"]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
T, Q, h, w = 8, 20, 90, 160
feats = torch.randn(T, 8, h, w, generator=g).to(dev)
params = (torch.randn(T * Q, 169, generator=g) * 0.3).to(dev)
refs = torch.rand(T * Q, 2, generator=g).to(dev)

x2 = torch.randn(7360, 384, generator=g).to(dev)
w2 = (torch.randn(1536, 384, generator=g) / 20).to(dev)
b2 = torch.randn(1536, generator=g).to(dev)
x0 = torch.randn(117760, 96, generator=g).to(dev)
w0 = (torch.randn(384, 96, generator=g) / 10).to(dev)
b0 = torch.randn(384, generator=g).to(dev)
qkv = torch.randn(1, 4, 48, 80, 3 * 192, generator=g).to(dev)        # stage-1 sized window attention
table = (torch.randn(2535, 6, generator=g) * 0.1).to(dev)
qkv_bias = torch.randn(3 * 192, generator=g).to(dev)


def k1(split):
    with hot_ops.use_matmul_mode('split' if split else 'f32'):
        return hot_ops.window_attention3d(qkv, qkv_bias, table, 6, (8, 7, 7), (4, 3, 3))


def make(mode):
    if mode == "k20":
        return lambda: hot_ops.linear_split(x2, w2, b2, act="gelu")
    if mode == "k20_s0":
        return lambda: hot_ops.linear_split(x0, w0, b0, act="gelu")
    if mode == "k1":
        return lambda: k1(True)
    if mode == "k1f32":
        return lambda: k1(False)
    if mode == "k13":                     # K13b (bf16 matrix cores) in the default mode, the f32-MFMA K13 under SOC_MATMUL=f32
        return lambda: hot_ops.ws_linear(x0, w0, b0, act="gelu")
    if mode == "lib":
        return lambda: torch.nn.functional.linear(x2, w2, b2)
    return lambda: None


side = torch.cuda.Stream()
first = hot_ops.dynamic_mask(feats, params, refs, (360.0, 640.0), 4).clone()
torch.cuda.synchronize()
res = {}
for mode in modes:
    big = make(mode)
    try:
        big()
    except Exception as e:  # noqa: BLE001
        res[mode] = f"unavailable: {e}"
        continue
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    lanes = torch.zeros(64, dtype=torch.int64, device=dev)
    halves = torch.zeros(2, dtype=torch.int64, device=dev)
    worst = torch.zeros((), device=dev)
    for i in range(N):
        for _ in range(4):
            big()
        with torch.cuda.stream(side):
            for _ in range(3):
                out = hot_ops.dynamic_mask(feats, params, refs, (360.0, 640.0), 4)
                ne = (out != first)
                bad += ne.any()
                idx = torch.nonzero(ne.view(T * Q, -1))[:, 1]
                lanes += torch.bincount((idx // 2) % 64, minlength=64)
                halves += torch.bincount(idx % 2, minlength=2)
                worst = torch.maximum(worst, (out - first).abs().max())
    torch.cuda.synchronize()
    res[mode] = {"k4_launches": 3 * N, "launches_with_wrong_elements": int(bad), "worst_abs": float(worst),
                 "by_half(even_x,odd_x)": halves.tolist(),
                 "by_lane_group(0-15,16-31,32-47,48-63)": lanes.view(4, 16).sum(1).tolist()}
    print(mode, json.dumps(res[mode]), flush=True)
print(json.dumps(res))
