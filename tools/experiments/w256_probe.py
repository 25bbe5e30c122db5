"""Round 6: the 256 x 256 linear layers at launch-group row counts (the encoder's value_proj / output_proj + shortcut: 385 600 rows;
the fusion blocks' projections: 288 000 / 73 600) on K13b (weights in LDS, three / four column ranges), K24 (rows split once,
weights streamed), K20 (tiles) and the f32 library GEMM."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurips2023_soc_amd import hot_ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


g = torch.Generator().manual_seed(0)
for rows, N, K, res in ((385600, 256, 256, False), (385600, 256, 256, True), (288000, 256, 256, False), (73600, 256, 256, False), (38560, 256, 256, False),
                        (385600, 384, 256, False)):
    x = torch.randn(rows, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(rows, N, generator=g).cuda() if res else None
    fl = 2e-6 * rows * N * K
    by = 4e-3 * (rows * K + rows * N * (2 if res else 1))
    line = f"rows {rows:6d} N {N} K {K} res {int(res)}:"
    for name, fn in (("library", lambda: (F.linear(x, w, b) + r) if res else F.linear(x, w, b)),
                     ("K13b", lambda: hot_ops.ws_linear(x, w, b, None, r)),
                     ("K24", lambda: hot_ops.xs_linear(x, w, b, None, r)),
                     ("K20", lambda: hot_ops.linear_split(x, w, b, None, r))):
        try:
            t = timeit(fn)
            line += f"  {name} {t:7.1f} us ({fl / t:5.1f} TF, {by / t:5.2f} TB/s)"
        except Exception as e:      # noqa: BLE001
            line += f"  {name} -- ({type(e).__name__})"
    print(line, flush=True)
