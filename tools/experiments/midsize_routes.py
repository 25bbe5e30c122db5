"""The mid-size linear layers K13b / K20 keep at the row counts of a four-clip launch group: K24 (with column spans), K13b, K20
and the library on the same shapes.
    python tools/experiments/midsize_routes.py"""
import sys
import torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops  # noqa: E402
g = torch.Generator().manual_seed(0)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); torch.cuda._sleep(20_000_000)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


for name, M, N, K, ln, res, act in [("s1.qkv", 115200, 576, 192, True, False, "none"), ("s1.proj", 115200, 192, 192, False, True, "none"),
                                    ("inproj1", 115200, 256, 192, False, False, "none"), ("vlf.q1", 115200, 256, 256, False, False, "none"),
                                    ("enc.value", 154240, 256, 256, False, False, "none"), ("enc.out", 154240, 256, 256, False, True, "none"),
                                    ("enc.offs", 154240, 384, 256, False, False, "none"), ("vlf.out2", 29440, 256, 256, False, False, "none")]:
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    lnp = ((torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda() * 0.1, 1e-5) if ln else None
    r = torch.randn(M, N, generator=g).cuda() if res else None
    fl = 2.0 * M * N * K
    by = 4.0 * (M * K + M * N * (2 if res else 1))
    out = []
    try:
        u = t(lambda: hot_ops.xs_linear(x, w, b, lnp, r, act)); out.append(f"K24 {u:6.1f} plan {hot_ops.xs_linear_plan(M, N, K)}")
    except Exception as e:
        out.append(f"K24 - ({type(e).__name__})")
    if hot_ops.ws_linear_supported(x, w, lnp):
        try:
            out.append(f"K13b {t(lambda: hot_ops.ws_linear(x, w, b, lnp, r, act)):6.1f}")
        except Exception as e:
            out.append(f"K13b - ({e})")
    try:
        out.append(f"K20 {t(lambda: hot_ops.linear_split(x, w, b, ln=lnp, residual=r, act=act)):6.1f}")
    except Exception as e:
        out.append(f"K20 - ({type(e).__name__})")
    print(f"{name:9s} {M}x{N}x{K}: floor max(MFMA {fl / 417e6:.0f}, HBM {by / 6.3e6:.0f}) us   " + "   ".join(out), flush=True)
