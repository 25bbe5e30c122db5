"""FPN spatial decoder alone (hipGraph replay) with MIOpen's default pick vs torch.backends.cudnn.benchmark (find mode),
NCHW vs channels_last inputs."""
import sys
import time

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402

model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
fpn = model.spatial_decoder.cuda().eval()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(8, 256, 12, 20, device="cuda", generator=g)
feats = [torch.randn(8, 256, 23, 40, device="cuda", generator=g), torch.randn(8, 256, 45, 80, device="cuda", generator=g),
         torch.randn(8, 96, 90, 160, device="cuda", generator=g)]


def timed(fn, reps=50):
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                out = fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fn()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out


t, ref = timed(lambda: fpn(x, feats))
print(f"default pick, NCHW          {t:.3f} ms", flush=True)
# the token-major ladder (K19 / K10 / K18): memory levels in place, feats0 channels-last
shapes = [(45, 80), (23, 40), (12, 20)]
memory = torch.randn(8, sum(h * w for h, w in shapes) + 60, 256, device="cuda", generator=g)
f0 = feats[2].contiguous(memory_format=torch.channels_last)
t, o = timed(lambda: fpn.forward_tokens(memory, shapes, f0))
print(f"token-major ladder          {t:.3f} ms", flush=True)
xc, fc = x.contiguous(memory_format=torch.channels_last), [f.contiguous(memory_format=torch.channels_last) for f in feats]
t, o = timed(lambda: fpn(xc, fc))
print(f"default pick, channels_last {t:.3f} ms   max|d| {(o - ref).abs().max().item():.2e}", flush=True)
torch.backends.cudnn.benchmark = True
t0 = time.perf_counter()
t, o = timed(lambda: fpn(x, feats))
print(f"find mode, NCHW             {t:.3f} ms   max|d| {(o - ref).abs().max().item():.2e}  (first-call cost {time.perf_counter() - t0:.1f} s)", flush=True)
t0 = time.perf_counter()
t, o = timed(lambda: fpn(xc, fc))
print(f"find mode, channels_last    {t:.3f} ms   max|d| {(o - ref).abs().max().item():.2e}  (first-call cost {time.perf_counter() - t0:.1f} s)", flush=True)
