"""Video-Swin stage 2 (C = 384) at the row counts of a launch group (4 clips: 29 440 rows; 2: 14 720; 1: 7 360): the layers of
the "k20" flow (LayerNorm folded into K20's qkv, K20 proj + shortcut, K23 plain) against the "k23" flow (K24 qkv without
LayerNorm, K24 proj + shortcut, K23 with norm1 of the next block as second output), HIP events around back-to-back launches."""
import sys

import torch

sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops  # noqa: E402

reps = 30
g = torch.Generator().manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 384


def t(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


for M in (7360, 14720, 29440):
    x = torch.randn(M, C, generator=g).cuda()
    wq, bq = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda(), torch.randn(3 * C, generator=g).cuda()
    wp, bp = (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), torch.randn(C, generator=g).cuda()
    w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).cuda(), torch.randn(4 * C, generator=g).cuda()
    w2, b2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).cuda(), torch.randn(C, generator=g).cuda()
    ln = ((torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda() * 0.1, 1e-5)
    stats = hot_ops.row_stats(x, 1e-5)
    r = {}
    r["k20 qkv + LN (incl. row stats)"] = t(lambda: hot_ops.linear_split(x, wq, bq, ln=ln, stats=hot_ops.row_stats(x, 1e-5)))
    r["k24 qkv"] = t(lambda: hot_ops.xs_linear(x, wq, bq, None, None, "none"))
    r["k20 proj + res"] = t(lambda: hot_ops.linear_split(x, wp, bp, None, x, "none"))
    r["k24 proj + res"] = t(lambda: hot_ops.xs_linear(x, wp, bp, None, x, "none"))
    r["k23 plain"] = t(lambda: hot_ops.mlp_split(x, w1, b1, w2, b2, "gelu", ln, x))
    r["k23 + norm1 of next"] = t(lambda: hot_ops.mlp_split(x, w1, b1, w2, b2, "gelu", ln, x, post_ln=ln, return_sum=True))
    a = r["k20 qkv + LN (incl. row stats)"] + r["k20 proj + res"] + r["k23 plain"]
    b = r["k24 qkv"] + r["k24 proj + res"] + r["k23 + norm1 of next"]
    print(f"M = {M}, C = {C}: " + ", ".join(f"{k} {v:.1f}" for k, v in r.items()) + f"  |  k20 flow {a:.1f} us, k23 flow {b:.1f} us per block")
