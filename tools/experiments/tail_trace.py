"""Replay only the tail of a launch group (B clips, VOC over independent clips) so that rocprofv3 --kernel-trace --stats lists
its kernels: python tools/experiments/tail_trace.py [B] [reps]"""
import sys

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import group_tail  # noqa: E402
from neurips2023_soc_amd.nested_tensor import NestedTensor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T, H, Wd, L = 8, 360, 640, 10
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clip = torch.stack([W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(B)], 1).contiguous()
pad = torch.zeros(T, B, H, Wd, dtype=torch.bool, device="cuda")
ids = torch.cat([W.synthetic_token_ids(1 + i, L) for i in range(B)], 0).cuda()
text = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
t1 = [[{"size": (H, Wd)}] for _ in range(T)]
rec = torch.zeros(B, 1 + T * 20 + T * 90 * 160, device="cuda")
with torch.no_grad():
    sb = model.forward_head(NestedTensor(clip, pad, unpadded=True), None, text)
    for _ in range(2):
        group_tail(model, sb, t1, False, rec)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        group_tail(model, sb, t1, False, rec)
    torch.cuda.synchronize()
    marker = torch.zeros(1, device="cuda")
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
print("done")
