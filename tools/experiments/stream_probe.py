#!/usr/bin/env python
"""Where the streamed (H2D-inclusive) form of bench.py loses time against the resident form: the H2D copy alone, the
pipelined replay alone, and the two together with different feeder depths / hand-over forms."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.clip_io import DoubleBufferedH2D  # noqa: E402
from neurips2023_soc_amd.graph_runner import PipelinedClipGraph  # noqa: E402

dev = torch.device("cuda")
T, H, Wd, L = 8, 360, 640, 10
N = int(os.environ.get('PROBE_N', 40))
NH = int(os.environ.get('PROBE_NHOST', 8))
model, _, _ = S.build_model(S.default_args("video-swin-t", text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.to(dev).eval()
host = [W.synthetic_clip(1 + i, T, H, Wd).pin_memory() for i in range(NH)]
ids = W.synthetic_token_ids(1, L).to(dev)
res = {}

# H2D alone
dst = torch.empty(T, 3, H, Wd, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    dst.copy_(host[i % NH], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
res["h2d_ms"] = 1e3 * dt
res["h2d_GBps"] = host[0].numel() * 4 / dt / 1e9

g = PipelinedClipGraph(model, T, H, Wd, L, dev)
clips = [h.to(dev) for h in host[:4]]


def run(feed_depth=None, release_early=False):
    feeder = DoubleBufferedH2D((T, 3, H, Wd), torch.float32, dev, depth=feed_depth) if feed_depth else None
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if feeder:
            feeder.submit(host[0])
        for i in range(N):
            if feeder:
                if i + 1 < N:
                    feeder.submit(host[(i + 1) % NH])
                c = feeder.acquire()
            else:
                c = clips[i % 4]
            if release_early and feeder:
                g.clip.copy_(c.view(g.clip.shape), non_blocking=True)
                feeder.release()
                g.steady[g._n % 2].replay()
                g._n += 1
            else:
                g.run(c, ids)
                if feeder:
                    feeder.release()
        g.flush()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
    return 1e3 * dt


res["resident_ms"] = run()
res["stream_depth2_ms"] = run(2)
res["stream_depth3_ms"] = run(3)
res["stream_depth2_release_after_d2d_ms"] = run(2, True)
res["stream_depth3_release_after_d2d_ms"] = run(3, True)
res["resident_again_ms"] = run()
print(json.dumps(res))
