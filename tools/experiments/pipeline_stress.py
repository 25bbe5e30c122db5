"""Stress: many replays of the two-stage PipelinedClipGraph (hang / drift check).  Run under `timeout`.
usage: python tools/experiments/pipeline_stress.py [backbone] [T H W] [replays]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import PipelinedClipGraph  # noqa: E402

bb = sys.argv[1] if len(sys.argv) > 1 else "video-swin-t"
T, H, Wd = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8, 360, 640)
N = int(sys.argv[5]) if len(sys.argv) > 5 else 1500
model, _, _ = S.build_model(S.default_args(bb, text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(3)]
ids = W.synthetic_token_ids(1, 10).cuda()
pg = PipelinedClipGraph(model, T, H, Wd, 10, "cuda")
first = None
t0 = time.perf_counter()
for r in range(N):
    rec = pg.run(clips[r % 3], ids)
    if rec is not None and (r - 1) % 3 == 0:
        if first is None:
            first = rec.clone()
        elif r % 300 == 1:
            torch.cuda.synchronize()
            print(f"replay {r}: drift {float((rec - first).abs().max()):.2e}, {1e3 * (time.perf_counter() - t0) / (r + 1):.2f} ms/clip",
                  flush=True)
pg.flush()
torch.cuda.synchronize()
print(f"{bb} T={T} {H}x{Wd}: {N} replays ok, {1e3 * (time.perf_counter() - t0) / N:.2f} ms per clip", flush=True)
