import sys, torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops, fused
g = torch.Generator().manual_seed(0)
M, F = 38560, 2048
x = torch.randn(M, 256, generator=g).cuda()
w1 = (torch.randn(F, 256, generator=g) / 16).cuda(); b1 = torch.randn(F, generator=g).cuda()
w2 = (torch.randn(256, F, generator=g) / 45).cuda(); b2 = torch.randn(256, generator=g).cuda()
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); torch.cuda._sleep(20_000_000)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps
k22 = t(lambda: hot_ops.ffn_split(x, w1, b1, w2, b2))
two = t(lambda: torch.nn.functional.linear(fused.linear(x, w1, b1, relu=True), w2, b2))
print(f"K22 {k22:.1f} us ({4.0 * M * F * 256 / k22 / 1e6:.1f} TFLOP/s)   linear1 + ReLU (K13b / K20) + linear2 (library): {two:.1f} us")
M1 = (M // (16 * 8 * 256)) * (16 * 8 * 256)
def hybrid():
    out = torch.empty_like(x)
    out[:M1] = hot_ops.ffn_split(x[:M1], w1, b1, w2, b2)
    out[M1:] = torch.nn.functional.linear(fused.linear(x[M1:], w1, b1, relu=True), w2, b2)
    return out
print(f"hybrid: K22 on {M1} rows + two GEMMs on {M - M1}: {t(hybrid):.1f} us;  K22 alone on {M1} rows: {t(lambda: hot_ops.ffn_split(x[:M1], w1, b1, w2, b2)):.1f} us; two GEMMs on the rest: {t(lambda: torch.nn.functional.linear(fused.linear(x[M1:], w1, b1, relu=True), w2, b2)):.1f} us")
