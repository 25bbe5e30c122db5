import sys
import time

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import synthetic_dataset  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.infer import ClipInferencer  # noqa: E402

model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
engine = ClipInferencer(model, "cuda", use_graphs=True)
tok = synthetic_dataset.HashTokenizer()
clip = torch.randn(8, 3, 360, 640, device="cuda")
ids = tok("a b c d e f g h").cuda()
bufs = [torch.empty((8, 720, 1280), dtype=torch.bool, pin_memory=True) for _ in range(2)]
frames = torch.zeros(8, 720, 1280, 3, dtype=torch.uint8).pin_memory()
for _ in range(3):
    engine(clip, ids, (720, 1280))


def loop(mode, n=40):
    torch.cuda.synchronize()
    prev = None
    t = time.perf_counter()
    for i in range(n):
        if "h2d" in mode and i % 3 == 0:
            frames.to("cuda", non_blocking=True)
        if "tokpin" in mode:
            ids2 = tok("a b c d e f g h").pin_memory().to("cuda", non_blocking=True)
        if "tokplain" in mode:
            ids2 = tok("a b c d e f g h").to("cuda")
        if "bigpin" in mode and i % 3 == 0:
            torch.empty(8, 720, 1280, 3, dtype=torch.uint8).pin_memory().to("cuda", non_blocking=True)
        if "k9" in mode and i % 3 == 0:
            pre(frames)
        if "alloc" in mode and i % 3 == 0:
            torch.empty(8 * 720 * 1280 * 3 + i * 4096, dtype=torch.uint8, device="cuda")
        m = engine(clip, ids, (720, 1280))["masks"]
        if "d2h" in mode:
            bufs[i % 2].copy_(m, non_blocking=True)
        if "event" in mode:
            ev = torch.cuda.Event()
            ev.record()
            if prev is not None:
                prev.synchronize()
            prev = ev
        if "numpy" in mode:
            bufs[(i + 1) % 2].numpy().copy()
    torch.cuda.synchronize()
    print(f"{mode:28s} {1e3 * (time.perf_counter() - t) / n:6.2f} ms/clip", flush=True)


from neurips2023_soc_amd import clip_io  # noqa: E402
pre = clip_io.FramePreprocessor("cuda")
pre(frames)
for mode in ("d2h+event+numpy", "tokpin+d2h+event", "tokplain+d2h+event", "bigpin+d2h+event", "k9+d2h+event",
             "alloc+d2h+event", "plain"):
    loop(mode)
