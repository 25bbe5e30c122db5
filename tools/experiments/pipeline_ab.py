"""Same process, same box: the one-graph pipeline (PipelinedClipGraph), the two-stream pipeline (TwoStreamClipGraph) and the
bare two-stream schedule of partition_probe.py (no staging, no hand-over copies), alternating, bench.py's loop.
usage: python tools/experiments/pipeline_ab.py [clips per pass] [passes]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import PipelinedClipGraph, TwoStreamClipGraph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T, H, Wd, L = 8, 360, 640, 10
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(4)]
ids = W.synthetic_token_ids(1, L).cuda()
pipes = {"one-graph": PipelinedClipGraph(model, T, H, Wd, L, "cuda"), "two-stream": TwoStreamClipGraph(model, T, H, Wd, L, "cuda")}
out = torch.zeros(n, pipes["one-graph"].record.numel(), device="cuda")


def run(pipe, stage=True, copy=True):
    done = 0
    for i in range(n):
        if stage:
            pipe.stage_inputs(clips[i % 4], ids)
        rec = pipe.replay()
        if rec is not None:
            if copy:
                out[done].copy_(rec, non_blocking=True)
            done += 1
    for rec in pipe.flush():
        out[done].copy_(rec, non_blocking=True)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def bare():
    """partition_probe.py's schedule on the product's graphs: no staging, no caller-stream waits, one state slot"""
    ts = pipes["two-stream"]
    main, aux = ts.main, ts.aux
    ev_text, ev_head = torch.cuda.Event(), torch.cuda.Event()
    ev_head.record(main)
    for i in range(n):
        with torch.cuda.stream(aux):
            ts.g_text[0].replay()
            ev_text.record(aux)
            aux.wait_event(ev_head)
            ts.g_tail[0].replay()
        with torch.cuda.stream(main):
            ts.g_video[0].replay()
            main.wait_event(ev_text)
            ts.g_fuse[0].replay()
            ev_head.record(main)


res = {}
for name, pipe in pipes.items():
    run(pipe)
for p in range(passes):
    for name, pipe in pipes.items():
        res.setdefault(name, []).append(round(timed(lambda: run(pipe)), 3))
    res.setdefault("bare two-stream schedule", []).append(round(timed(bare), 3))
    res.setdefault("two-stream, no staging", []).append(round(timed(lambda: run(pipes["two-stream"], stage=False)), 3))
    res.setdefault("two-stream, no record copy", []).append(round(timed(lambda: run(pipes["two-stream"], copy=False)), 3))
# host time of one period's enqueue (no GPU wait): is the host ahead?
pipe = pipes["two-stream"]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    pipe.stage_inputs(clips[i % 4], ids)
    pipe.replay()
host = 1e3 * (time.perf_counter() - t0) / n
pipe.flush()
torch.cuda.synchronize()
res["two-stream host enqueue ms per clip"] = round(host, 3)
pipe = pipes["one-graph"]
t0 = time.perf_counter()
for i in range(n):
    pipe.stage_inputs(clips[i % 4], ids)
    pipe.replay()
res["one-graph host enqueue ms per clip"] = round(1e3 * (time.perf_counter() - t0) / n, 3)
pipe.flush()
torch.cuda.synchronize()
print(json.dumps(res, indent=1))
