import sys, torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops
g = torch.Generator(device="cuda").manual_seed(0)
M, K, N = 115200, 96, 288
x = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
b = torch.randn(N, device="cuda", generator=g)
gam, bet = torch.rand(K, device="cuda", generator=g) + 0.5, torch.randn(K, device="cuda", generator=g) * 0.1
for _ in range(12):
    hot_ops.ws_linear(x, w, b, (gam, bet, 1e-5), None, "none")
for _ in range(12):
    hot_ops.ws_linear(x, w, None, None, None, "none")
torch.cuda.synchronize()
