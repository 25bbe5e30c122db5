"""Runs the token-major FPN ladder a few times (for rocprofv3 --kernel-trace --stats)."""
import sys

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402

model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
fpn = model.spatial_decoder.cuda().eval()
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(45, 80), (23, 40), (12, 20)]
memory = torch.randn(8, sum(h * w for h, w in shapes) + 60, 256, device="cuda", generator=g)
f0 = torch.randn(8, 96, 90, 160, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        fpn.forward_tokens(memory, shapes, f0)
torch.cuda.synchronize()
