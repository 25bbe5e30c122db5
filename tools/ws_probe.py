"""K13 (weight-stationary linear) vs the library GEMM (+ the K5 / GELU passes it replaces) on the Video-Swin stage-0/1
layers of the BASELINE config.  usage: python tools/ws_probe.py"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from neurips2023_soc_amd import gemm_tuning, hot_ops  # noqa: E402

gemm_tuning.enable_tuned_gemms()


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for tag, M, K, N, ln, res, act in [
        ("s0 norm1+qkv", 115200, 96, 288, True, False, "none"), ("s0 proj+res", 115200, 96, 96, False, True, "none"),
        ("s0 norm2+fc1+gelu", 115200, 96, 384, True, False, "gelu"), ("s0 fc2+res", 115200, 384, 96, False, True, "none"),
        ("s1 norm1+qkv", 28800, 192, 576, True, False, "none"), ("s1 proj+res", 28800, 192, 192, False, True, "none"),
        ("s1 norm2+fc1+gelu", 28800, 192, 768, True, False, "gelu")]:
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    gam, bet = torch.rand(K, device="cuda", generator=g) + 0.5, torch.randn(K, device="cuda", generator=g) * 0.1
    r = torch.randn(M, N, device="cuda", generator=g) if res else None

    def lib():
        h = F.layer_norm(x, (K,), gam, bet, 1e-5) if ln else x
        y = F.linear(h, w, b)
        if act == "gelu":
            y = F.gelu(y)
        return y + r if res else y

    def mine():
        return hot_ops.ws_linear(x, w, b, (gam, bet, 1e-5) if ln else None, r, act)
    err = float((mine() - lib()).abs().max())
    t_lib, t_mine = timeit(lib), timeit(mine)
    t_gemm = timeit(lambda: F.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"{tag:20s} M={M:6d} K={K:3d} N={N:3d}: K13 {t_mine:6.1f} us ({fl / t_mine / 1e6:5.1f} TF)   library GEMM alone "
          f"{t_gemm:6.1f} us, with its LN/GELU/add passes {t_lib:6.1f} us   max|diff| {err:.1e}", flush=True)
