"""What does one more (tiny / mid-size) launch in the tail cost the pipelined replay?  Adds K dummy dependent launches to the
tail and reports the change of the per-clip time."""
import sys
import time

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import PipelinedClipGraph  # noqa: E402

T, H, Wd, L = 8, 360, 640, 10
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clip = W.synthetic_clip(1, T, H, Wd).cuda()
ids = W.synthetic_token_ids(1, L).cuda()
tiny = torch.zeros(64, device="cuda")
mid = torch.zeros(160 * 256 * 16, device="cuda")         # 160 workgroups of 256 threads x float4 x 4
orig_tail = model.forward_tail
extra = {"n": 0, "buf": tiny}


def tail_plus(state, targets, fork=True):
    out = orig_tail(state, targets, fork=fork)
    for _ in range(extra["n"]):
        extra["buf"].add_(1.0)
    return out


model.forward_tail = tail_plus


def run(n=150):
    with torch.no_grad():
        g = PipelinedClipGraph(model, T, H, Wd, L, "cuda")
        for _ in range(6):
            g.run(clip, ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.run(clip, ids)
        g.flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


base = run()
print(f"baseline                     {base:.3f} ms/clip", flush=True)
for name, buf in (("tiny (1 workgroup)", tiny), ("mid (160 x 256 threads)", mid)):
    extra["buf"] = buf
    for k in (100, 300):
        extra["n"] = k
        t = run()
        print(f"+{k:3d} {name:24s} {t:.3f} ms/clip  -> {(t - base) / k * 1e3:6.2f} us per launch", flush=True)
extra["n"] = 0
print(f"baseline again               {run():.3f} ms/clip")
