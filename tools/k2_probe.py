"""K2 (multi-scale deformable attention, fused form) at the encoder size of the BASELINE configurations.
usage: python tools/k2_probe.py [reps] [360p|720p] [offset scale] [clips]      (run under rocprofv3 --pmc ... for the cache counters)
clips: frames = 8 x clips in one launch (a launch group: 10 -> N = 80)
360p: S = 4 820 per frame (configs 1-3, 5); 720p: S = 19 160 (config 4)."""
import sys

import torch

sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops  # noqa: E402
from neurips2023_soc_amd.deformable_transformer import DeformableTransformerEncoder as E  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
which = sys.argv[2] if len(sys.argv) > 2 else "360p"
shapes = {"360p": [[45, 80], [23, 40], [12, 20], [6, 10]], "720p": [[90, 160], [45, 80], [23, 40], [12, 20]]}[which]
CLIPS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N, M = 8 * CLIPS, 8
sh = torch.tensor(shapes).cuda()
lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
S = int(sh.prod(1).sum())
g = torch.Generator().manual_seed(0)
value = torch.randn(N, S, M, 32, generator=g).cuda()
ref = E.get_reference_points(shapes, torch.ones(N, 4, 2), "cpu").cuda()
# offsets like the model's: a per-(head, level, point) pattern of 1-4 px plus a query-dependent part (std ~2 px)
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
pattern = torch.randn(1, 1, M, 4, 4, 2, generator=g) * 2.0 * scale
off = (pattern + torch.randn(N, S, M, 4, 4, 2, generator=g) * 1.6 * scale).cuda()
print(f"offset scale {scale}: mean |off| {float(off.abs().mean()):.2f} px, P(|off| > 5) = {float((off.abs() > 5).float().mean()):.3f}")
logits = torch.randn(N, S, M, 16, generator=g).cuda()
algo = (value.numel() + off.numel() + logits.numel() + N * S * M * 32) * 4


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


us = timeit(lambda: hot_ops.msda_fused_forward(value, sh, lsi, ref, off, logits))
taps = N * S * M * 64                       # 16 points x 4 bilinear taps per (query, head), 128 B (two 64-B L1 accesses) each
print(f"K2 fused {which} S={S}: {us:7.1f} us  {algo / us / 1e3:7.1f} GB/s algorithmic ({algo / us / 1e3 / 8000:.3f} of HBM peak); "
      f"gather {taps * 128 / 1e9:.2f} GB through the L1s = {taps * 128 / us / 1e6:.1f} TB/s, {2 * taps / 1e6:.1f} M 64-B accesses")
