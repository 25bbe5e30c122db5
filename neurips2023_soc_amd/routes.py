"""Which kernel runs each pixel-sized linear layer of a forward, from the clip geometry alone.

The dispatch rules live where they are applied (`fused.route_linear / route_gelu / route_linear_multi / mlp_ok`,
`video_swin.BasicLayer.stage_flow`); they only look at shapes, dtypes and devices.  This module walks the layers of a
configuration with stand-in tensors and returns {call site: route}, so that

* `tests/test_routes.py` (CPU) pins the table of the BASELINE configurations (`tests/golden/routes.json`): a threshold that
  silently sends a layer back to the library GEMM fails a test instead of showing up as milliseconds;
* the same test on the GPU box checks that the launches a real forward records are the ones the table names.

    python -m neurips2023_soc_amd.routes --write      (re-generate the golden after a deliberate change)
"""
from __future__ import annotations

import json
import math
import os
import sys
from types import SimpleNamespace
from typing import Dict, Tuple

import torch

from . import fused, hot_ops
from .video_swin import SWIN_CONFIGS, WINDOW, BasicLayer

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "routes.json")
# BASELINE.json `configs`: Swin-T 360 x 640 (configs 0-2), Swin-B 720 x 1280 (3), Swin-B 360 x 640 (4); T = 8 throughout
BASELINE_GEOMETRIES = (("video-swin-t", 8, 360, 640), ("video-swin-b", 8, 720, 1280), ("video-swin-b", 8, 360, 640))
D_MODEL, D_FFN = 256, 2048


class Stand:
    """Stand-in for a CUDA float32 tensor: the routing predicates read shape / dtype / device only."""
    is_cuda = True
    dtype = torch.float32

    def __init__(self, *shape: int):
        self.shape = tuple(shape)

    def numel(self) -> int:
        return math.prod(self.shape)


def _lin(n: int, k: int, bias: bool = True):
    return SimpleNamespace(weight=Stand(n, k), bias=Stand(n) if bias else None)


def table(backbone: str, T: int, H: int, W: int, clips: int = 1) -> Dict[str, str]:
    """`clips` independent clips per launch group (graph_runner.QuadPipelinedClipGraph: 4): a batch of `clips`, so every
    pixel-sized layer sees `clips` times the rows."""
    cfg = SWIN_CONFIGS[backbone]
    T = T * clips                  # the routing predicates read row counts: frames of all clips of the group
    d = cfg["embed_dim"]
    h, w = -(-H // 4), -(-W // 4)
    out: Dict[str, str] = {}
    sizes = []
    for s in range(4):
        C = d * 2 ** s
        rows = T * h * w
        sizes.append((C, h, w))
        x = Stand(1, T, h, w, C)
        layer = BasicLayer.__new__(BasicLayer)          # the decision methods only read blocks[0]'s layer shapes
        blk = SimpleNamespace(attn=SimpleNamespace(qkv=_lin(3 * C, C), proj=_lin(C, C)),
                              mlp=SimpleNamespace(fc1=_lin(4 * C, C), fc2=_lin(C, 4 * C)))
        object.__setattr__(layer, "_modules", {})
        layer.__dict__["blocks"] = [blk]
        flow = BasicLayer.stage_flow(layer, x)
        out[f"swin{s}.flow"] = flow
        hid = Stand(1, T, h, w, 4 * C)
        if flow == "ws":
            out[f"swin{s}.qkv"] = out[f"swin{s}.proj"] = "k13b"
            if fused.mlp_ok(x, blk.mlp.fc1, blk.mlp.fc2):
                out[f"swin{s}.mlp"] = "k23"
            else:
                out[f"swin{s}.fc1"] = "k13b"
                out[f"swin{s}.fc2"] = "k13b" if hot_ops.ws_linear_supported(hid, blk.mlp.fc2.weight, False) else "library"
        elif flow == "k20":
            out[f"swin{s}.qkv"] = out[f"swin{s}.proj"] = "k20"
            if fused.mlp_ok(x, blk.mlp.fc1, blk.mlp.fc2):
                out[f"swin{s}.mlp"] = "k23"
            else:
                out[f"swin{s}.fc1"] = "k20"
                out[f"swin{s}.fc2"] = fused.route_linear(hid, blk.mlp.fc2.weight, blk.mlp.fc2.bias, residual=x)
        elif flow == "k24":
            out[f"swin{s}.qkv"] = out[f"swin{s}.proj"] = out[f"swin{s}.fc1"] = "k24"
            out[f"swin{s}.fc2"] = fused.route_linear(hid, blk.mlp.fc2.weight, blk.mlp.fc2.bias, residual=x)
        else:
            out[f"swin{s}.qkv"] = fused.route_linear(x, blk.attn.qkv.weight, blk.attn.qkv.bias)
            out[f"swin{s}.proj"] = fused.route_linear(x, blk.attn.proj.weight, blk.attn.proj.bias, residual=x)
            if flow == "k23":
                out[f"swin{s}.mlp"] = "k23"
            else:
                out[f"swin{s}.fc1"] = fused.route_gelu(x, blk.mlp.fc1.weight)
                out[f"swin{s}.fc2"] = "library"
        if s < 3:
            h, w = -(-h // 2), -(-w // 2)
            merged = Stand(1, T, h, w, 4 * C)
            out[f"merge{s}"] = fused.route_linear(merged, Stand(2 * C, 4 * C), None)
    # input_proj of levels 1..3 (1x1 convolutions as GEMMs over tokens), fusion blocks, deformable encoder
    enc_rows = 0
    for lvl in (1, 2, 3):
        C, hh, ww = sizes[lvl]
        tok = Stand(T, hh * ww, C)
        out[f"input_proj{lvl}"] = fused.route_linear(tok, Stand(D_MODEL, C), Stand(D_MODEL))
        seq = Stand(T * hh * ww, 1, D_MODEL)
        out[f"vlf{lvl}.q"] = fused.route_linear(seq, Stand(D_MODEL, D_MODEL), Stand(D_MODEL), add=None)
        out[f"vlf{lvl}.out*tgt"] = fused.route_linear(seq, Stand(D_MODEL, D_MODEL), Stand(D_MODEL), mul=seq)
        enc_rows += hh * ww
    _, hh, ww = sizes[3]
    enc_rows += -(-hh // 2) * -(-ww // 2)                  # the extra stride-64 level (3x3 stride-2 convolution: MIOpen)
    mem = Stand(T, enc_rows, D_MODEL)
    out["encoder.value_proj"] = fused.route_linear(mem, Stand(D_MODEL, D_MODEL), Stand(D_MODEL))
    out["encoder.offsets|weights"] = fused.route_linear_multi(
        mem, [(Stand(256, D_MODEL), Stand(256), True), (Stand(128, D_MODEL), Stand(128), True)], mem)
    out["encoder.output_proj"] = fused.route_linear(mem, Stand(D_MODEL, D_MODEL), Stand(D_MODEL))
    out["encoder.ffn"] = "k23" if fused.mlp_ok(mem, _lin(D_FFN, D_MODEL), _lin(D_MODEL, D_FFN)) else "library"
    return out


# what bench.py runs by default: groups of ten clips up to 360 x 640 (round 5: fours / eights), pairs above
GROUPED = (("video-swin-t", 8, 360, 640, 4), ("video-swin-b", 8, 360, 640, 4), ("video-swin-b", 8, 720, 1280, 2),
           ("video-swin-t", 8, 360, 640, 8), ("video-swin-b", 8, 360, 640, 8),
           ("video-swin-t", 8, 360, 640, 10), ("video-swin-b", 8, 360, 640, 10))


def all_tables() -> Dict[str, Dict[str, str]]:
    tabs = {f"{b} T={t} {hh}x{ww}": table(b, t, hh, ww) for b, t, hh, ww in BASELINE_GEOMETRIES}
    tabs.update({f"{b} T={t} {hh}x{ww} x{c} clips": table(b, t, hh, ww, c) for b, t, hh, ww, c in GROUPED})
    return tabs


if __name__ == "__main__":
    tabs = all_tables()
    if "--write" in sys.argv:
        with open(GOLDEN, "w") as f:
            json.dump(tabs, f, indent=1, sort_keys=True)
            f.write("\n")
    print(json.dumps(tabs, indent=1))
