"""K7 (linear_small) against the library GEMM at the row counts the query chain reaches in large launch groups (160 frame
queries x clips): where should fused.is_small() stop?   python tools/experiments/small_rows.py"""
import sys
import torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops  # noqa: E402
g = torch.Generator().manual_seed(0)


def t(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


for N, K in ((256, 256), (512, 256), (2048, 256), (256, 2048), (768, 768), (3072, 768), (768, 3072)):
    row = []
    for M in (100, 640, 1280, 1600, 1920, 2560, 3200, 4096):
        x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
        k7 = t(lambda: hot_ops.linear_small(x, w, b))
        lib = t(lambda: torch.nn.functional.linear(x, w, b))
        row.append(f"M={M}: K7 {k7:5.1f} lib {lib:5.1f}")
    print(f"N={N} K={K}:  " + "   ".join(row), flush=True)
