"""K24 (xs_linear_split.hip) against K13b / K20 / the library on the model's shapes."""
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


for name, M, N, K, ln, res, act in [("s1.qkv", 28800, 576, 192, True, False, "none"), ("s1.proj", 28800, 192, 192, False, True, "none"),
                                    ("s2.qkv", 7360, 1152, 384, False, False, "none"), ("s2.proj", 7360, 384, 384, False, True, "none"),
                                    ("merge1", 7360, 384, 768, False, False, "none"), ("s3.qkv", 1920, 2304, 768, True, False, "none"),
                                    ("s3.proj", 1920, 768, 768, False, True, "none"), ("s3.fc1", 1920, 3072, 768, True, False, "gelu"),
                                    ("enc.value", 38560, 256, 256, False, False, "none"), ("enc.out", 38560, 256, 256, False, True, "none"),
                                    ("vlf.q", 28800, 256, 256, False, False, "none"), ("inproj1", 28800, 256, 192, False, False, "none"),
                                    ("inproj2", 7360, 256, 384, False, False, "none")]:
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    lnp = ((torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda() * 0.1, 1e-5) if ln else None
    r = torch.randn(M, N, generator=g).cuda() if res else None
    fl = 2.0 * M * N * K
    k24 = t(lambda: hot_ops.xs_linear(x, w, b, lnp, r, act))
    plan = hot_ops.xs_linear_plan(M, N, K)
    others = []
    if hot_ops.ws_linear_supported(x, w, ln):
        try:
            others.append(f"K13b {t(lambda: hot_ops.ws_linear(x, w, b, lnp, r, act)):.1f}")
        except Exception:
            pass
    try:
        others.append(f"K20 {t(lambda: hot_ops.linear_split(x, w, b, ln=lnp, residual=r, act=act)):.1f}")
    except Exception:
        pass

    def lib():
        xx = torch.nn.functional.layer_norm(x, (K,), lnp[0], lnp[1], 1e-5) if lnp else x
        y = torch.nn.functional.linear(xx, w, b)
        y = torch.nn.functional.gelu(y) if act == "gelu" else y
        return y if r is None else y + r
    others.append(f"library(+passes) {t(lib):.1f}")
    alt = ""
    for cut in [(plan[0], plan[1] * 2), (plan[0], max(1, plan[1] // 2)), (min(plan[0] * 2, (M + 15) // 16), plan[1])]:
        try:
            alt += f"  cut{cut} {t(lambda: hot_ops.xs_linear(x, w, b, lnp, r, act, cut=cut)):.1f}"
        except Exception:
            pass
    print(f"{name:9s} {M}x{N}x{K}: K24 {k24:6.1f} us ({fl / k24 / 1e6:5.1f} TFLOP/s) plan {plan}   " + "  ".join(others) + alt, flush=True)
