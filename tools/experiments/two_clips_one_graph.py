"""Experiment: TWO independent clip forwards captured as parallel branches of ONE hipGraph (the earlier attempt --
two graphs replayed concurrently on two streams -- hung, tools/experiments/README.md).
Usage (GPU box, under a short `timeout`): python tools/experiments/two_clips_one_graph.py [T H W] [n_inflight]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import postprocessing as P, weights as W  # noqa: E402

T, H, Wd = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 360, 640)
NF = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda")
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.to(dev).eval()
clips = [torch.zeros(T, 1, 3, H, Wd, device=dev) for _ in range(NF)]
pad = torch.zeros(T, 1, H, Wd, dtype=torch.bool, device=dev)
ids = W.synthetic_token_ids(1, 10).to(dev)
text = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
targets = [[{"size": (H, Wd)}] for _ in range(T)]
for i, c in enumerate(clips):
    c.copy_(W.synthetic_clip(1 + i, T, H, Wd).to(dev)[:, None])


def fwd(i):
    out = model(S.NestedTensor(clips[i], pad, unpadded=True), None, text, targets)
    return P.select_trajectory(out)[1]


branch = [torch.cuda.Stream(device=dev) for _ in range(NF - 1)]
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(2):
        ref = [fwd(i).clone() for i in range(NF)]
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager ok", flush=True)

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g), torch.no_grad():
    cur = torch.cuda.current_stream()
    outs = [None] * NF
    for i, s in enumerate(branch):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs[i + 1] = fwd(i + 1)
    outs[0] = fwd(0)
    for s in branch:
        cur.wait_stream(s)
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed once; max diff vs eager:", [float((o - r).abs().max()) for o, r in zip(outs, ref)], flush=True)
for n in (5, 20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{n} replays of {NF} clips: {1e3 * dt / (n * NF):.2f} ms per clip, {n * NF / dt:.1f} clips/s", flush=True)
