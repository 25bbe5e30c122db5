"""Where does the dataset driver's host time go?  Replays infer_refytb's loop on a synthetic dataset with a
timer around every host step.  Usage: python tools/e2e_probe.py ROOT"""
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import clip_io, infer_refytb, synthetic_dataset  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.infer import ClipInferencer  # noqa: E402

root = sys.argv[1]
synthetic_dataset.make_dataset(root, videos=16, frames=8, expressions=3, n_words=8)
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
engine = ClipInferencer(model, "cuda", use_graphs=True)
tok = synthetic_dataset.HashTokenizer()
img_folder, data = infer_refytb.load_meta(root)
todo = sorted(data)
acc = defaultdict(float)


class timer:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t = time.perf_counter()

    def __exit__(self, *a):
        acc[self.name] += time.perf_counter() - self.t


def one_pass():
    cache = clip_io.VideoClipCache(clip_io.FramePreprocessor("cuda"), workers=8)
    bufs = [torch.empty((8, 720, 1280), dtype=torch.bool, pin_memory=True) for _ in range(2)]
    prev, n = None, 0
    for vi, video in enumerate(todo):
        frames = data[video]["frames"]
        paths = clip_io.frame_paths(img_folder, video, frames)
        if vi + 1 < len(todo):
            with timer("prefetch_submit"):
                cache.prefetch(clip_io.frame_paths(img_folder, todo[vi + 1], data[todo[vi + 1]]["frames"]))
        for exp_id, item in data[video]["expressions"].items():
            with timer("cache_get"):
                clip, orig = cache.get(paths)
            with timer("tokenize"):
                ids = tok(item["exp"]).pin_memory().to("cuda", non_blocking=True)
            with timer("engine_enqueue"):
                masks = engine(clip, ids, orig)["masks"]
            with timer("d2h_enqueue"):
                host = bufs[n % 2]
                host.copy_(masks, non_blocking=True)
                done = torch.cuda.Event()
                done.record()
            if prev is not None:
                with timer("event_wait"):
                    prev[1].synchronize()
                with timer("numpy_copy"):
                    prev[0].numpy().copy()
            prev = (host, done)
            n += 1
    prev[1].synchronize()
    return n


for rep in range(3):
    acc.clear()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = one_pass()
    torch.cuda.synchronize()
    total = time.perf_counter() - t
    print(f"pass {rep}: {1e3 * total / n:.2f} ms/clip;", {k: round(1e3 * v / n, 2) for k, v in acc.items()}, flush=True)
