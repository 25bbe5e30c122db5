"""Round 6: the long-K layers still on the library (swin3.fc2: K = 3072 -> 768 + shortcut; merge2: K = 1536 -> 768) on K20 (hot_ops.linear_split)
against torch's f32 GEMM, at one clip and at a launch group of ten."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurips2023_soc_amd import gemm_tuning, hot_ops  # noqa: E402


def timeit(fn, reps=30):
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


gemm_tuning.load_tuning() if hasattr(gemm_tuning, "load_tuning") else None
g = torch.Generator().manual_seed(0)
for rows, N, K, res in ((1920, 768, 3072, True), (19200, 768, 3072, True), (7680, 768, 3072, True), (1920, 768, 1536, False),
                        (19200, 768, 1536, False), (7680, 768, 1536, False), (19200, 256, 256, False), (2400, 256, 768, False), (24000, 256, 768, False)):
    x = torch.randn(rows, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(rows, N, generator=g).cuda() if res else None
    ref = (x.double() @ w.double().t() + b.double() + (r.double() if res else 0)).float()
    lib = lambda: F.linear(x, w, b) + r if res else F.linear(x, w, b)      # noqa: E731
    t_lib = timeit(lib)
    e_lib = float((lib() - ref).abs().max() / ref.abs().max())
    line = f"rows {rows:6d} N {N:4d} K {K:4d} res {int(res)}: library {t_lib:7.1f} us ({2e-6 * rows * N * K / t_lib:6.1f} TFLOP/s, err {e_lib:.1e})"
    if hot_ops.linear_split_supported(x, w):
        k20 = lambda: hot_ops.linear_split(x, w, b, None, r)         # noqa: E731
        t = timeit(k20)
        err = float((k20() - ref).abs().max() / ref.abs().max())
        line += f"   K20 {t:7.1f} us ({2e-6 * rows * N * K / t:6.1f}, err {err:.1e})"
    else:
        line += "   K20 unsupported"
    print(line)
