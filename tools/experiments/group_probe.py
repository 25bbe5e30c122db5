"""Replay times of the pieces of a launch group (B clips per launch, VOC per clip): text || Video-Swin, fusion + encoder, head,
tail -- per group and per clip.  usage: python tools/experiments/group_probe.py [B ...]"""
import sys

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import group_tail  # noqa: E402
from neurips2023_soc_amd.nested_tensor import NestedTensor  # noqa: E402

T, H, Wd, L = 8, 360, 640, 10
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
reps = 20


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def time_ms(g):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for B in [int(v) for v in sys.argv[1:]] or [1, 2, 4]:
    clip = torch.stack([W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(B)], 1).contiguous()
    pad = torch.zeros(T, B, H, Wd, dtype=torch.bool, device="cuda")
    ids = torch.cat([W.synthetic_token_ids(1 + i, L) for i in range(B)], 0).cuda()
    text = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    t1 = [[{"size": (H, Wd)}] for _ in range(T)]
    rec = torch.zeros(B, 1 + T * 20 + T * 90 * 160, device="cuda")
    with torch.no_grad():
        g_b, sa = capture(lambda: model.forward_backbone(NestedTensor(clip.clone(), pad.clone(), unpadded=True), None, text))
        g_v, _ = capture(lambda: model.forward_video(NestedTensor(clip.clone(), pad.clone(), unpadded=True)))
        g_x, _ = capture(lambda: model.forward_text_state(text, ids.device))
        g_f, sb = capture(lambda: model.forward_fuse_encode(sa))
        g_t, _ = capture(lambda: group_tail(model, sb, t1, True, rec))
        g_t1, _ = capture(lambda: group_tail(model, sb, t1, False, rec))
        r = {"text || swin": time_ms(g_b), "swin": time_ms(g_v), "text": time_ms(g_x), "fuse + encoder": time_ms(g_f),
             "tail (forked)": time_ms(g_t), "tail (one stream)": time_ms(g_t1)}
    print(f"B = {B}: " + ", ".join(f"{k} {v:.3f} ({v / B:.3f} per clip)" for k, v in r.items()), flush=True)
    del g_b, g_v, g_x, g_f, g_t, g_t1
