"""Round-6 debugging aid: special inputs through the streaming K1 form (shifted), error in frames 0-3 / 4-7."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurips2023_soc_amd import hot_ops as ops  # noqa: E402
from oracle import soc_oracle as O  # noqa: E402


def run(tag, qkv, bias, table, nH, shift):
    ref = O.window_attention_core(qkv, bias, table, nH, O.WINDOW, shift)
    new = ops.window_attention3d(qkv.cuda(), bias.cuda(), table.cuda(), nH, O.WINDOW, shift).cpu()
    e = (new - ref).abs()[0]
    D = e.shape[0]
    print(f"{tag:34s} err frames 0-3 {float(e[:4].max()):.2e}   frames 4-{D - 1} {float(e[4:].max()):.2e}   per frame "
          + " ".join(f"{float(e[z].max()):.0e}" for z in range(D)))


g = torch.Generator().manual_seed(3)
nH, D, H, W = 1, 8, 14, 14
C = 32
qkv = torch.randn(1, D, H, W, 3 * C, generator=g)
bias = torch.randn(3 * C, generator=g) * 0.5
table = torch.randn(15 * 13 * 13, nH, generator=g) * 0.5
for shift in ((0, 0, 0), (4, 3, 3), (0, 3, 0), (0, 0, 3), (4, 0, 0)):
    run(f"random, shift {shift}", qkv, bias, table, nH, shift)
shift = (4, 3, 3)
q0 = qkv.clone(); q0[..., :C] = 0
run("Q = 0", q0, torch.cat([torch.zeros(C), bias[C:]]), table, nH, shift)
run("table = 0", qkv, bias, torch.zeros_like(table), nH, shift)
v1 = qkv.clone(); v1[..., 2 * C:] = 1.0
run("V = 1", v1, torch.cat([bias[:2 * C], torch.ones(C)]), table, nH, shift)
k0 = qkv.clone(); k0[..., C:2 * C] = 0
run("K = 0", k0, torch.cat([bias[:C], torch.zeros(C), bias[2 * C:]]), table, nH, shift)
run("K = 0, table = 0 (uniform P)", k0, torch.cat([bias[:C], torch.zeros(C), bias[2 * C:]]), torch.zeros_like(table), nH, shift)
qkv16 = torch.randn(1, 16, H, W, 3 * C, generator=g)
run("D = 16, shift (4,3,3)", qkv16, bias, table, nH, shift)
