"""Per-kernel cost of a chain of dependent kernels inside a hipGraph replay (launch gap + drain / ramp), for a tiny kernel
and for a chip-filling one."""
import time

import torch


def replay_ms(fn, n, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(n):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


x = torch.zeros(64, device="cuda")
big = torch.zeros(64 * 1024 * 1024, device="cuda")          # 256 MB: ~0.1 ms per pass
a = torch.randn(4096, 4096, device="cuda")
for name, fn in (("tiny add (64 floats)", lambda: x.add_(1.0)), ("256-MB add", lambda: big.add_(1.0)),
                 ("4096^3 f32 GEMM", lambda: torch.mm(a, a))):
    t1, t2 = replay_ms(fn, 50), replay_ms(fn, 250)
    print(f"{name:22s}: {(t2 - t1) / 200 * 1e3:8.2f} us per additional dependent launch (50 -> 250 launches: {t1:.3f} -> {t2:.3f} ms)")
