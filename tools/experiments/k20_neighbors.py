#!/usr/bin/env python
"""Does a running K20 launch disturb OTHER kernels on the chip (or the reverse)?  Main stream: K20 launches back to back.
Side stream: small LDS-using kernels of this package (K7 small linear, K3 attention core, K19 3x3 convolution, K5 LayerNorm).
Every output of both streams is compared bit for bit with its first value."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neurips2023_soc_amd import hot_ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator().manual_seed(3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
mode = sys.argv[2] if len(sys.argv) > 2 else "k20"

# main-stream work
M, K, Nn = 38560, 256, 2048
x = torch.randn(M, K, generator=g).to(dev)
w = (torch.randn(Nn, K, generator=g) / 16).to(dev)
b = torch.randn(Nn, generator=g).to(dev)
big = {"k20": lambda: hot_ops.linear_split(x, w, b, act="relu"),
       "lib": lambda: torch.relu(torch.nn.functional.linear(x, w, b)),
       "k1": None}[mode] if mode != "k1" else None
if mode == "k1":
    qkv = torch.randn(1, 8, 90, 160, 288, generator=g).to(dev)
    qb = torch.randn(288, generator=g).to(dev)
    tab = (torch.randn(2535, 3, generator=g) * 0.2).to(dev)
    big = lambda: hot_ops.window_attention3d(qkv, qb, tab, 3, (8, 7, 7), (4, 3, 3))   # noqa: E731

# side-stream work
xs = torch.randn(160, 256, generator=g).to(dev)
ws = (torch.randn(256, 256, generator=g) / 16).to(dev)
bs = torch.randn(256, generator=g).to(dev)
q3, k3, v3 = (torch.randn(L, 1, 256, generator=g).to(dev) for L in (160, 160, 160))
cx = torch.randn(8, 45 * 80, 128, generator=g).to(dev)
cw = (torch.randn(64, 9 * 128, generator=g) / 30).to(dev)
lnx = torch.randn(4820, 256, generator=g).to(dev)
lnw, lnb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
small = [("K7 linear_small", lambda: hot_ops.linear_small(xs, ws, bs, None, True)),
         ("K3 mha_core", lambda: hot_ops.mha_core(q3, k3, v3, 8, None)),
         ("K19 conv3x3_tokens", lambda: hot_ops.conv3x3_tokens(cx, (45, 80), cw, None)),
         ("K5 add_layernorm", lambda: hot_ops.add_layernorm(lnx, lnx, lnw, lnb, 1e-5, return_sum=False)[1])]

side = torch.cuda.Stream()
first_big = big()
first_small = [f() for _, f in small]
torch.cuda.synchronize()
bad_big = torch.zeros((), dtype=torch.int64, device=dev)
bad_small = [torch.zeros((), dtype=torch.int64, device=dev) for _ in small]
for i in range(N):
    y = big()
    bad_big += (y != first_big).any()
    with torch.cuda.stream(side):
        for j, (_, f) in enumerate(small):
            z = f()
            bad_small[j] += (z != first_small[j]).any()
torch.cuda.synchronize()
print(json.dumps({"main": mode, "launches": N, "main_differing": int(bad_big),
                  **{name: int(bs_) for (name, _), bs_ in zip(small, bad_small)}}))
