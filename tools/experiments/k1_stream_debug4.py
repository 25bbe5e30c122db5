"""Round-6 debugging aid: raw O and row sums of the pipelined tile with P == 1 (K = 0, table = 0, V = 1): both must be 392."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k1_stream_debug3 import build, call  # noqa: E402

nH, D, H, W = 1, 8, 7, 7
Cc = 32
g = torch.Generator().manual_seed(3)
qkv = torch.randn(1, D, H, W, 3 * Cc, generator=g)
qkv[..., Cc:2 * Cc] = 0
qkv[..., 2 * Cc:] = 1
bias = torch.zeros(3 * Cc); bias[2 * Cc:] = 1
table = torch.zeros(15 * 13 * 13, nH)
for tag, flag in (("raw O", "-DSOC_K1_DBG=3"), ("row sum", "-DSOC_K1_DBG=4")):
    lib = build([flag], "d" + flag[-1])
    out = call(lib, qkv.cuda(), bias.cuda(), table.cuda(), nH, (0, 0, 0))[0]          # [D, 7, 7, 32]
    print(tag, "per (dz) min / max over the window and dims:")
    for z in range(D):
        print(f"   dz {z}: {float(out[z].min()):.3f} .. {float(out[z].max()):.3f}   dims of token (0,0): "
              + " ".join(f"{float(v):.0f}" for v in out[z, 0, 0, :8]))
