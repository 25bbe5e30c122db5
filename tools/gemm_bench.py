"""Measure torch/hipBLASLt fp32 GEMM throughput on the hot path's shapes (decision input)."""
import torch
import torch.nn.functional as F

shapes = [  # (name, M, K, N)
    ("s0 qkv", 115200, 96, 288), ("s0 proj", 115200, 96, 96), ("s0 fc1", 115200, 96, 384), ("s0 fc2", 115200, 384, 96),
    ("s1 qkv", 28800, 192, 576), ("s1 fc1", 28800, 192, 768), ("s1 fc2", 28800, 768, 192),
    ("s2 qkv", 7360, 384, 1152), ("s2 fc1", 7360, 384, 1536), ("s2 fc2", 7360, 1536, 384),
    ("s3 qkv", 1920, 768, 2304), ("s3 fc1", 1920, 768, 3072), ("s3 fc2", 1920, 3072, 768),
    ("enc lin1", 38560, 256, 2048), ("enc lin2", 38560, 2048, 256), ("enc vproj", 38560, 256, 256),
    ("enc off+w", 38560, 256, 384), ("vlf in", 28800, 256, 256),
]
for name, M, K, N in shapes:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        F.linear(x, w, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        F.linear(x, w, b)
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 10
    fl = 2.0 * M * K * N
    by = 4.0 * (M * K + K * N + M * N)
    print(f"{name:10s} M={M:6d} K={K:4d} N={N:4d}  {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s  {by / us / 1e3:7.1f} GB/s")
