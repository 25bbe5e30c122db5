import sys, torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops
g = torch.Generator(device="cuda").manual_seed(0)
for T, h, w in ((8, 90, 160), (8, 180, 320), (36, 90, 160)):
    feats = torch.randn(T, 8, h, w, device="cuda", generator=g)
    params = torch.randn(T * 20, 169, device="cuda", generator=g) * 0.3
    refs = torch.rand(T * 20, 2, device="cuda", generator=g)
    for _ in range(3):
        hot_ops.dynamic_mask(feats, params, refs, (4 * h, 4 * w), 4)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        hot_ops.dynamic_mask(feats, params, refs, (4 * h, 4 * w), 4)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    mb = (feats.numel() + params.numel() + T * 20 * h * w) * 4 / 1e6
    print(f"T={T} {h}x{w}: {us:.1f} us, {mb / us * 1e3 / 1e3:.2f} TB/s algorithmic")
