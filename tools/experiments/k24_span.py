"""K24 column spans (round 5): a workgroup keeps its split rows for several column ranges.  Times every legal cut
(workgroup rows x column spans) of the model's K24 shapes at the row counts of one clip and of a launch group of four, checks
that all cuts give the same bits, and prints the library's plan.
    python tools/experiments/k24_span.py [reps [clips ...]]"""
import sys
import torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops  # noqa: E402
g = torch.Generator().manual_seed(0)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CUS = 256


def t(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); torch.cuda._sleep(20_000_000)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


SHAPES = [("s2.qkv", 7360, 1152, 384, True, False, "none"), ("s2.proj", 7360, 384, 384, False, True, "none"),
          ("merge1", 7360, 384, 768, True, False, "none"), ("s3.qkv", 1920, 2304, 768, True, False, "none"),
          ("s3.proj", 1920, 768, 768, False, True, "none"), ("s3.fc1", 1920, 3072, 768, True, False, "gelu"),
          ("merge0", 28800, 192, 384, True, False, "none"), ("inproj2", 7360, 256, 384, False, False, "none"),
          ("vlf.q2", 7360, 256, 256, False, False, "none")]
for clips in ([int(v) for v in sys.argv[2:]] or [4, 1]):
    for name, M1, N, K, ln, res, act in SHAPES:
        M = M1 * clips
        x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
        lnp = ((torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda() * 0.1, 1e-5) if ln else None
        r = torch.randn(M, N, generator=g).cuda() if res else None
        fl = 2.0 * M * N * K
        plan = hot_ops.xs_linear_plan(M, N, K)
        ref = hot_ops.xs_linear(x, w, b, lnp, r, act)
        t_plan = t(lambda: hot_ops.xs_linear(x, w, b, lnp, r, act))
        nw = 4 if K > 384 else 8
        groups = ((M + 15) // 16 + nw - 1) // nw
        rows = []
        for ncr in range(1, N // 16 + 1):
            if (N // 16) % ncr:
                continue
            for nrg in sorted({min(groups, max(1, CUS // ncr)), min(groups, max(1, 2 * CUS // ncr)), groups}):
                try:
                    out = hot_ops.xs_linear(x, w, b, lnp, r, act, cut=(nrg, ncr))
                except Exception:
                    continue
                same = bool(torch.equal(out, ref))
                rows.append((t(lambda: hot_ops.xs_linear(x, w, b, lnp, r, act, cut=(nrg, ncr))), nrg, ncr, same))
        rows.sort()
        best = rows[0]
        print(f"x{clips} {name:8s} {M}x{N}x{K}: plan {plan} {t_plan:6.1f} us ({fl / t_plan / 1e6:5.1f} TF)   best cut ({best[1]},{best[2]}) {best[0]:6.1f} us"
              f"   all cuts same bits: {all(r_[3] for r_ in rows)}   " + "  ".join(f"({a},{c}) {u:.0f}" for u, a, c, _ in rows[:6]), flush=True)
