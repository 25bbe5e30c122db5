#!/usr/bin/env python
"""Which K20 call site makes the pipelined replay deviate?  N back-to-back replays at the BASELINE size; every record is
compared with the first one of its clip.  Run with SOC_MATMUL=f32 or SOC_SPLIT_OFF=swin,gelu,relu,multi,mul,res,add,plain."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import PipelinedClipGraph  # noqa: E402

T, H, Wd, L = 8, 360, 640, 10
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(3)]
ids = W.synthetic_token_ids(1, L).cuda()
pg = PipelinedClipGraph(model, T, H, Wd, L, "cuda")
first, devs = {}, []
for r in range(N):
    rec = pg.run(clips[r % 3], ids)
    if rec is not None:
        k = (r - 1) % 3
        if k not in first:
            first[k] = rec.clone()
        else:
            devs.append((rec - first[k]).abs().max())
pg.flush()
torch.cuda.synchronize()
d = torch.stack(devs).cpu()
print(json.dumps({"matmul": os.environ.get("SOC_MATMUL", "split"), "off": os.environ.get("SOC_SPLIT_OFF", ""), "replays": N,
                  "deviating_records(>2e-4)": int((d > 2e-4).sum()), "worst": float(d.max()),
                  "median": float(d.median())}))
