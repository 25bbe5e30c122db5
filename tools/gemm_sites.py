"""Every large linear layer of one clip forward, by kernel and shape, with its GPU time: which GEMM sites matter and how
far each is from the matrix-core rate.  Serial eager forward (side streams off), one HIP-event pair per call with the host
kept ahead by a spin kernel.  Usage: python tools/gemm_sites.py [reps]"""
import collections
import sys

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import hot_ops, weights as W  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T, H, Wd = 8, 360, 640
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
model._side_stream = lambda device: None
clip = W.synthetic_clip(1, T, H, Wd).cuda()
ids = W.synthetic_token_ids(1, 10).cuda()
calls = []


def wrap(mod, name, shape_of):
    fn = getattr(mod, name)

    def inner(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **k)
        e.record()
        calls.append((name, shape_of(*a, **k), s, e))
        return out
    setattr(mod, name, inner)


def lin_shape(x, weight, *a, **k):
    Kd = x.shape[-1]
    extra = []
    ln = k.get("ln", a[1] if len(a) > 1 else None)
    if ln is not None:
        extra.append("ln")
    for key in ("act", "residual", "add", "mul"):
        v = k.get(key)
        if v is not None and v != "none":
            extra.append(key if key != "act" else str(v))
    return (x.numel() // Kd, weight.shape[0], Kd, "+".join(extra))


wrap(hot_ops, "linear_split", lin_shape)
wrap(hot_ops, "ws_linear", lin_shape)
wrap(hot_ops, "linear_act", lambda x, w, *a, **k: (x.numel() // x.shape[-1], w.shape[0], x.shape[-1], str(a[1] if len(a) > 1 else k.get("act", ""))))
_F = torch.nn.functional
_lin = _F.linear


def f_linear(x, w, b=None):
    if x.is_cuda and x.numel() // x.shape[-1] >= 2048:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = _lin(x, w, b)
        e.record()
        calls.append(("library", (x.numel() // x.shape[-1], w.shape[0], x.shape[-1], ""), s, e))
        return out
    return _lin(x, w, b)


_F.linear = f_linear
agg = collections.OrderedDict()
with torch.no_grad():
    for r in range(reps + 1):
        del calls[:]
        torch.cuda._sleep(200_000_000)
        samples = S.NestedTensor(clip[:, None], torch.zeros(T, 1, H, Wd, dtype=torch.bool, device="cuda"), unpadded=True)
        model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}, [[{"size": (H, Wd)}]] * T)
        torch.cuda.synchronize()
        if r == 0:
            continue
        for name, shp, s, e in calls:
            agg.setdefault((name, shp), []).append(s.elapsed_time(e))
tot = collections.Counter()
print(f"{'kernel':14s} {'M':>7s} {'N':>5s} {'K':>5s} {'epilogue':18s} {'calls':>5s} {'us/call':>8s} {'TFLOP/s':>8s} {'ms/clip':>8s}")
for (name, (M, N, K, extra)), ts in agg.items():
    n = len(ts) // reps
    us = 1e3 * sum(ts) / len(ts)
    tot[name] += us * n / 1e3
    print(f"{name:14s} {M:7d} {N:5d} {K:5d} {extra:18s} {n:5d} {us:8.1f} {2.0 * M * N * K / us / 1e6:8.1f} {us * n / 1e3:8.3f}")
print({k: round(v, 3) for k, v in tot.items()})
