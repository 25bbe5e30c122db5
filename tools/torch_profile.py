"""torch.profiler view of one eager forward: which aten ops (with shapes) own the non-GEMM GPU time."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402

model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
T, H, Wd = 8, 360, 640
clip = W.synthetic_clip(1, T, H, Wd).cuda()
ids = W.synthetic_token_ids(1, 10).cuda()


def fwd():
    samples = S.NestedTensor(clip[:, None], torch.zeros(T, 1, H, Wd, dtype=torch.bool, device="cuda"))
    return model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}, [[{"size": (H, Wd)}]] * T)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    fwd()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=300,
                                                         max_name_column_width=28, max_shapes_column_width=70))
