"""Encoder projections as ONE library GEMM: src [38560,256] x [W_v; W_off; W_att]^T (N = 640) + a cached constant matrix
(value bias | pos W_off^T + b_off | pos W_att^T + b_att) through addmm(beta = 1), against today's value GEMM + K12."""
import os
import sys
import time

import torch

M, K, N = 38560, 256, 640
g = torch.Generator(device="cuda").manual_seed(0)
src = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) * 0.05
Cm = torch.randn(M, N, device="cuda", generator=g)
bias = torch.randn(N, device="cuda", generator=g)


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"tunable: enabled={os.environ.get('PYTORCH_TUNABLEOP_ENABLED')} tuning={os.environ.get('PYTORCH_TUNABLEOP_TUNING')}")
print(f"addmm(C, src, W^T)  N=640         {t(lambda: torch.addmm(Cm, src, W.t())):7.1f} us")
print(f"linear(src, W, bias) N=640        {t(lambda: torch.nn.functional.linear(src, W, bias)):7.1f} us")
print(f"addmm N=384 (offsets | weights)   {t(lambda: torch.addmm(Cm[:, :384].contiguous(), src, W[:384].t())):7.1f} us (incl. a 59-MB copy)")
C384 = Cm[:, :384].contiguous()
print(f"addmm N=384, C contiguous         {t(lambda: torch.addmm(C384, src, W[:384].t())):7.1f} us")
print(f"linear N=256 (value_proj)         {t(lambda: torch.nn.functional.linear(src, W[:256], bias[:256])):7.1f} us")
out = torch.empty(M, N, device="cuda")
print(f"addmm out= N=640                  {t(lambda: torch.addmm(Cm, src, W.t(), out=out)):7.1f} us")
