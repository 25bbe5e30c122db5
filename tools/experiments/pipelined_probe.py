"""PipelinedClipGraph vs ClipGraph: results and throughput.  GPU box, under a short timeout.
usage: python tools/experiments/pipelined_probe.py [T H W]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import ClipGraph, PipelinedClipGraph  # noqa: E402

T, H, Wd = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 360, 640)
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(4)]
ids = W.synthetic_token_ids(1, 10).cuda()
g = ClipGraph(model, T, H, Wd, 10, "cuda")
ref = []
for c in clips:
    g.run(c, ids)
    ref.append(g.record.clone())
print("plain graph ok", flush=True)
DEPTH = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pg = PipelinedClipGraph(model, T, H, Wd, 10, "cuda")
print(f"pipelined graphs captured (depth {DEPTH})", flush=True)
got = []
for c in clips:
    r = pg.run(c, ids)
    if r is not None:
        got.append(r.clone())
got += pg.flush()
torch.cuda.synchronize()
print("max |record diff| per clip:", [float((a - b).abs().max()) for a, b in zip(got, ref)], len(got), flush=True)
for name, fn, fl in (("plain", lambda c: g.run(c, ids), None), ("pipelined", lambda c: pg.run(c, ids), pg.flush)):
    for n in (8, 40):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(clips[i % 4])
        if fl:
            fl()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"{name:10s} {n:3d} clips: {1e3 * dt / n:.2f} ms per clip, {n / dt:.1f} clips/s", flush=True)
