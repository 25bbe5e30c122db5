"""Experiment driver: hand-written fp32 MFMA GEMM (tools/microbench/gemm_f32.hip) vs the library GEMM on the
Swin shapes.  Usage (on the GPU box): python tools/gemm_probe.py"""
import ctypes as C
import os
import subprocess
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from neurips2023_soc_amd import gemm_tuning  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "microbench", "libgemm_probe.so")
src = os.path.join(here, "microbench", "gemm_f32.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", src, "-o", so],
                   check=True)
gemm_tuning.enable_tuned_gemms()
lib = C.CDLL(so)
lib.probe_gemm_nt.restype = C.c_int
lib.probe_gemm_nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]


def timeit(fn, n=20):
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


shapes = [(115200, 96, 288), (115200, 96, 96), (115200, 96, 384), (115200, 384, 96), (28800, 192, 576), (28800, 192, 768),
          (28800, 768, 192), (7360, 384, 1152), (7360, 384, 1536), (7360, 1536, 384), (1920, 768, 3072), (38560, 256, 2048),
          (38560, 2048, 256), (38560, 256, 256)]
for M, K, N in shapes:
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def mine(act=0, bk=16):
        rc = lib.probe_gemm_nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, act, bk, st)
        assert rc == 0, rc
    mine(0)
    ref = F.linear(a, w, b)
    err = float((out - ref).abs().max())
    mine(2)
    err_g = float((out - F.gelu(ref)).abs().max())
    t_lib = timeit(lambda: F.linear(a, w, b))
    t_lib_g = timeit(lambda: F.gelu(F.linear(a, w, b)))
    t_mine = timeit(lambda: mine(0))
    t_mine_g = timeit(lambda: mine(2))
    t32 = timeit(lambda: mine(0, 32))
    mine(0, 32)
    err32 = float((out - ref).abs().max())
    fl = 2.0 * M * N * K
    print(f"M={M:6d} K={K:4d} N={N:4d}  lib {t_lib:7.1f} us ({fl / t_lib / 1e6:5.1f} TF)  mine {t_mine:7.1f} us "
          f"({fl / t_mine / 1e6:5.1f} TF) bk32 {t32:7.1f} ({fl / t32 / 1e6:5.1f} TF) | +gelu lib {t_lib_g:7.1f}  mine {t_mine_g:7.1f}  "
          f"err {err:.1e} {err_g:.1e} {err32:.1e}", flush=True)
