import sys, torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops
g = torch.Generator().manual_seed(0)
M0 = 115200
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); torch.cuda._sleep(20_000_000)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps
for name, M, K, N, ln, res, act in [("qkv", M0, 96, 288, True, False, "none"), ("proj", M0, 96, 96, False, True, "none"),
                                    ("fc1", M0, 96, 384, True, False, "gelu"), ("fc2", M0, 384, 96, False, True, "none"),
                                    ("qkvB", M0, 128, 384, True, False, "none"), ("fc1B", M0, 128, 512, True, False, "gelu"),
                                    ("s1.qkv", 28800, 192, 576, True, False, "none"), ("s1.proj", 28800, 192, 192, False, True, "none"),
                                    ("s1.fc1", 28800, 192, 768, True, False, "gelu"),
                                    ("s2.qkv", 7360, 384, 1152, False, False, "none"), ("s2.proj", 7360, 384, 384, False, True, "none"),
                                    ("s2.fc1", 7360, 384, 1536, False, False, "gelu"),
                                    ("enc.value", 38560, 256, 256, False, False, "none"), ("enc.out", 38560, 256, 256, False, True, "none"),
                                    ("enc.offw", 38560, 256, 384, False, False, "none"), ("enc.ffn1", 38560, 256, 2048, False, False, "relu"),
                                    ("vlf.q", 28800, 256, 256, False, False, "none")]:
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    lnp = ((torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda() * 0.1, 1e-5) if ln else None
    r = torch.randn(M, N, generator=g).cuda() if res else None
    out = {}
    for mode in ("split", "f32"):
        try:
            with hot_ops.use_matmul_mode(mode):
                out[mode] = t(lambda: hot_ops.ws_linear(x, w, b, lnp, r, act))
        except Exception:                      # the f32 form has no LayerNorm at K = 384
            out[mode] = float("nan")
    fl = 2.0 * M * N * K
    print(f"{name:5s} {M}x{N}x{K}: K13b {out['split']:.1f} us ({fl / out['split'] / 1e6:.1f} TFLOP/s)   K13 {out['f32']:.1f} us ({fl / out['f32'] / 1e6:.1f})")
