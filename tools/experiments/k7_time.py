"""K7 on the text encoder's four linear layers (10 tokens) and two decoder shapes, back to back inside a graph replay:
time per launch.   python tools/experiments/k7_time.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from neurips2023_soc_amd import hot_ops  # noqa: E402

g = torch.Generator().manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shapes = [("qkv", 2304, 768, False), ("proj", 768, 768, False), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, False),
          ("dec 256", 256, 256, False), ("dec up", 2048, 256, True)]
for name, N, K, act in shapes:
    x = torch.randn(M, K, generator=g).cuda()
    ws = [(torch.randn(N, K, generator=g) / K ** 0.5).cuda() for _ in range(8)]     # 8 weight sets: no L2 reuse of one matrix
    b = torch.randn(N, generator=g).cuda()

    def chain():
        for w in ws:
            hot_ops.linear_small(x, w, b, None, act)
    chain()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(5):
            chain()
    gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    gr.replay()
    e.record()
    torch.cuda.synchronize()
    print(f"M={M} {name:8s} {N}x{K}: {s.elapsed_time(e) * 1e3 / 40:6.2f} us per launch", flush=True)
