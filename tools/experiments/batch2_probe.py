"""Two clips per forward (B = 2) against one (B = 1): hipGraph replay time of the whole forward + selection, per clip, and
the largest difference between the batched and the single-clip outputs.  usage: python tools/experiments/batch2_probe.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.nested_tensor import NestedTensor  # noqa: E402

T, H, Wd, L = 8, 360, 640, 10
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(2)]
ids = [W.synthetic_token_ids(1 + i, L).cuda() for i in range(2)]
res = {}
outs = {}
for B in (1, 2):
    clip = torch.stack(clips[:B], 1).contiguous()            # [T, B, 3, H, W]
    pad = torch.zeros(T, B, H, Wd, dtype=torch.bool, device="cuda")
    tok = torch.cat(ids[:B], 0)
    text = {"input_ids": tok, "attention_mask": torch.ones_like(tok)}
    targets = [[{"size": (H, Wd)}] * B for _ in range(T)]

    def fwd():
        return model(NestedTensor(clip.clone(), pad, unpadded=True), None, text, targets)
    for name, fn in (("forward", fwd), ("head", lambda: model.forward_head(NestedTensor(clip.clone(), pad, unpadded=True), None, text))):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(2):
                out = fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        res[f"B={B} {name} ms per clip"] = a.elapsed_time(b) / 20 / B
        if name == "forward":
            outs[B] = {k: out[k].clone() for k in ("pred_masks", "pred_cls")}
    del g
d = {k: float((outs[2][k][:, :1] - outs[1][k]).abs().max()) for k in outs[1]}
res["B=2 forward vs single, clip 0: max abs diff (the reference's tail couples the batch)"] = d

# head over two clips, tail per clip
clip = torch.stack(clips, 1).contiguous()
pad = torch.zeros(T, 2, H, Wd, dtype=torch.bool, device="cuda")
tok = torch.cat(ids, 0)
text = {"input_ids": tok, "attention_mask": torch.ones_like(tok)}
t1 = [[{"size": (H, Wd)}] for _ in range(T)]


def head2_tails():
    sb = model.forward_head(NestedTensor(clip.clone(), pad, unpadded=True), None, text)
    return [model.forward_tail(st, t1) for st in model.split_state(sb)]


singles = []
for b in range(2):
    c1 = clips[b][:, None].contiguous()
    singles.append(model(NestedTensor(c1, pad[:, :1], unpadded=True), None,
                         {"input_ids": ids[b], "attention_mask": torch.ones_like(ids[b])}, t1))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(2):
        both = head2_tails()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    both = head2_tails()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    g.replay()
b_.record()
torch.cuda.synchronize()
res["head B=2 + two B=1 tails: ms per clip"] = a.elapsed_time(b_) / 20 / 2
res["head B=2 + B=1 tails vs single forwards: max abs diff"] = {
    k: max(float((both[i][k] - singles[i][k]).abs().max()) for i in range(2)) for k in ("pred_masks", "pred_cls", "pred_boxes")}
print(json.dumps(res, indent=1))
