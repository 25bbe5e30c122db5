"""Kernel sequence (name, duration, gap) of the latency-bound query chain: decoder -> VOC -> heads."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402

model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
model._side_stream = lambda device: None
T, H, Wd = 8, 360, 640
clip = W.synthetic_clip(1, T, H, Wd).cuda()
ids = W.synthetic_token_ids(1, 10).cuda()
cap = {}
dec = model.transformer.decode
model.transformer.decode = lambda *a, **k: cap.setdefault("dec", (a, k)) and dec(*a, **k)
enc = model.transformer.encode
model.transformer.encode = lambda *a, **k: cap.setdefault("enc", (a, k)) and enc(*a, **k)
model.voc.register_forward_pre_hook(lambda m, i: cap.__setitem__("voc", i))


def fwd():
    samples = S.NestedTensor(clip[:, None], torch.zeros(T, 1, H, Wd, dtype=torch.bool, device="cuda"), unpadded=True)
    return model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}, [[{"size": (H, Wd)}]] * T)


with torch.no_grad():
    fwd()
    a, k = cap["dec"]
    which = sys.argv[1] if len(sys.argv) > 1 else "dec"

    def region():
        if which == "dec":
            return dec(*a, **k)
        if which == "enc":
            return enc(*cap["enc"][0], **cap["enc"][1])
        return model.voc(*cap["voc"])
    for _ in range(3):
        region()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        torch.cuda._sleep(50_000_000)
        region()
        torch.cuda.synchronize()
ev = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "spin" not in e.name),
            key=lambda e: e.time_range.start)
t0 = ev[0].time_range.start
prev_end = t0
for e in ev:
    print(f"{(e.time_range.start - t0):9.1f} us  gap {(e.time_range.start - prev_end):6.1f}  dur {e.time_range.elapsed_us():7.1f}  {e.name[:90]}")
    prev_end = e.time_range.end
print("launches", len(ev), "span us", ev[-1].time_range.end - t0)
