"""Which engine moves a pinned host clip to the device (a shader blit kernel or the SDMA engine), and what it costs a
whole-CU kernel that runs beside it.   python tools/experiments/h2d_engine.py   (env variations from the shell)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from neurips2023_soc_amd import hot_ops  # noqa: E402

print({k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "ROC", "GPU_", "AMD_"))})
dev = torch.device("cuda")
host = [torch.randn(8, 3, 360, 640).pin_memory() for _ in range(4)]
dst = [torch.empty(8, 3, 360, 640, device=dev) for _ in range(2)]
cs = torch.cuda.Stream()
g = torch.Generator().manual_seed(0)
M, C, F = 32768, 256, 2048
x = torch.randn(M, C, generator=g).cuda()
w1, b1 = (torch.randn(F, C, generator=g) / 16).cuda(), torch.randn(F, generator=g).cuda()
w2, b2 = (torch.randn(C, F, generator=g) / 45).cuda(), torch.randn(C, generator=g).cuda()


def work(n=20):
    for _ in range(n):
        hot_ops.mlp_split(x, w1, b1, w2, b2, "relu")


for h in host:
    dst[0].copy_(h, non_blocking=True)
work(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    dst[i % 2].copy_(host[i % 4], non_blocking=True)
torch.cuda.synchronize()
print(f"H2D alone: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per 22-MB clip")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); work(); e.record(); torch.cuda.synchronize()
alone = s.elapsed_time(e) / 20
s.record()
for i in range(20):
    with torch.cuda.stream(cs):
        dst[i % 2].copy_(host[i % 4], non_blocking=True)
    work(1)
e.record(); torch.cuda.synchronize()
print(f"K23 (256 whole-CU workgroups, {alone * 1e3:.0f} us alone): {s.elapsed_time(e) / 20 * 1e3:.0f} us per launch with an H2D copy beside each")

# the same with the work on a side stream (no legacy default stream involved), and with a library GEMM as the work
ws = torch.cuda.Stream()
a, b = torch.randn(8192, 4096, device=dev), torch.randn(4096, 4096, device=dev)
for name, fn in (("K23 on a side stream", lambda: work(1)), ("torch.mm 8192x4096x4096 on a side stream", lambda: torch.mm(a, b))):
    with torch.cuda.stream(ws):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s.record(ws)
        for _ in range(20):
            fn()
        e.record(ws)
        torch.cuda.synchronize()
        alone = s.elapsed_time(e) / 20
        s.record(ws)
        for i in range(20):
            with torch.cuda.stream(cs):
                dst[i % 2].copy_(host[i % 4], non_blocking=True)
            fn()
        e.record(ws)
        torch.cuda.synchronize()
        both = s.elapsed_time(e) / 20
        # one copy for every FOUR launches
        s.record(ws)
        for i in range(20):
            if i % 4 == 0:
                with torch.cuda.stream(cs):
                    dst[i % 2].copy_(host[i % 4], non_blocking=True)
            fn()
        e.record(ws)
        torch.cuda.synchronize()
    print(f"{name}: {alone * 1e3:.0f} us alone, {both * 1e3:.0f} us with a copy beside each, {s.elapsed_time(e) / 20 * 1e3:.0f} us with a copy per four launches")
