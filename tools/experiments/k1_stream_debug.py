"""Round-6 debugging aid: the streaming K1 form against the round-3 split form (SOC_K1_FORM=r3) and the oracle, per window."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurips2023_soc_amd import hot_ops as ops  # noqa: E402
from oracle import soc_oracle as O  # noqa: E402


def case(D, H, W, nH, shift, seed):
    g = torch.Generator().manual_seed(seed)
    C = nH * 32
    qkv = torch.randn(1, D, H, W, 3 * C, generator=g)
    bias = torch.randn(3 * C, generator=g) * 0.5
    table = torch.randn(15 * 13 * 13, nH, generator=g) * 0.5
    ref = O.window_attention_core(qkv, bias, table, nH, O.WINDOW, shift)
    os.environ["SOC_K1_FORM"] = ""
    new = ops.window_attention3d(qkv.cuda(), bias.cuda(), table.cuda(), nH, O.WINDOW, shift).cpu()
    os.environ["SOC_K1_FORM"] = "r3"
    old = ops.window_attention3d(qkv.cuda(), bias.cuda(), table.cuda(), nH, O.WINDOW, shift).cpu()
    os.environ["SOC_K1_FORM"] = ""
    print(f"D{D} H{H} W{W} nH{nH} shift{shift}: new vs ref {float((new - ref).abs().max()):.3e}  old vs ref {float((old - ref).abs().max()):.3e}")
    err = (new - ref).abs()[0]                    # [D, H, W, C]
    if float(err.max()) > 1e-4:
        e_tok = err.view(D, H, W, nH, 32).amax(-1)          # per token and head
        bad = (e_tok > 1e-4)
        print("  bad tokens per head:", bad.sum((0, 1, 2)).tolist(), "of", D * H * W)
        idx = bad.nonzero()
        print("  first bad (z, y, x, head):", idx[:12].tolist())
        print("  bad per frame z:", bad.sum((1, 2, 3)).tolist())
        print("  bad per row y:", bad.sum((0, 2, 3)).tolist())
        print("  bad per col x:", bad.sum((0, 1, 3)).tolist())


if __name__ == "__main__":
    case(8, 14, 21, 3, (0, 0, 0), 814)
    case(8, 14, 21, 3, (4, 3, 3), 814)
    case(8, 12, 20, 2, (0, 0, 0), 812)
    case(8, 12, 20, 6, (4, 3, 3), 812)
    case(8, 23, 40, 12, (4, 3, 3), 23)


def per_window(D, H, W, nH, shift, seed):
    g = torch.Generator().manual_seed(seed)
    C = nH * 32
    qkv = torch.randn(1, D, H, W, 3 * C, generator=g)
    bias = torch.randn(3 * C, generator=g) * 0.5
    table = torch.randn(15 * 13 * 13, nH, generator=g) * 0.5
    ref = O.window_attention_core(qkv, bias, table, nH, O.WINDOW, shift)
    new = ops.window_attention3d(qkv.cuda(), bias.cuda(), table.cuda(), nH, O.WINDOW, shift).cpu()
    err = (new - ref).abs()[0].amax(-1)                     # [D, H, W]
    sh = (0 if D <= 8 else shift[0], shift[1], shift[2])
    e = torch.roll(err, shifts=(-sh[0], -sh[1], -sh[2]), dims=(0, 1, 2))      # shifted frame: windows are aligned blocks
    print(f"per-window max error, D{D} H{H} W{W} shift{shift} (rows wy, cols wx), frames 0-3 | 4-7:")
    for wy in range(H // 7):
        row = []
        for wx in range(W // 7):
            blk = e[:, wy * 7:wy * 7 + 7, wx * 7:wx * 7 + 7]
            row.append(f"{float(blk[:4].max()):.1e}|{float(blk[4:].max()):.1e}")
        print("   ", "  ".join(row))
    blk = e[:, 0:7, 0:7]
    print("    window (0,0), max error per (dy, dx) over dz<4:")
    for dy in range(7):
        print("      ", " ".join(f"{float(blk[:4, dy, dx].max()):.0e}" for dx in range(7)))


if __name__ == "__main__":
    per_window(8, 28, 35, 1, (4, 3, 3), 5)
