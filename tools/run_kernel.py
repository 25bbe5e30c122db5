"""Launch one hot kernel repeatedly at a BASELINE-config geometry (for rocprofv3 --pmc passes).
usage: python tools/run_kernel.py {k1s0|k1s1|k1s2|k1s3|msda|vlf|dyn|k20ffn|k20qkv1|k20fc1s2|k13qkv0|k13fc1s0|k13qkv2|k23s0|k23s1|k23s2|k23enc} [reps]
(SOC_MATMUL=f32 in the environment keeps K1 on the f32-input MFMA form: the ablation partner of the default split form)"""
import sys

import torch

sys.path.insert(0, ".")
from neurips2023_soc_amd import hot_ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "k1s0"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
CLIPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # clips per launch group (bench.py's default pipeline: 4): rows x CLIPS
g = torch.Generator().manual_seed(0)
dev = "cuda"
if which.startswith("k1s"):
    st = int(which[-1])
    H, W, nH = [(90, 160, 3), (45, 80, 6), (23, 40, 12), (12, 20, 24)][st]
    C = nH * 32
    qkv = torch.randn(CLIPS, 8, H, W, 3 * C, generator=g).to(dev)
    bias = torch.randn(3 * C, generator=g).to(dev)
    table = (torch.randn(2535, nH, generator=g) * 0.2).to(dev)
    fn = lambda: hot_ops.window_attention3d(qkv, bias, table, nH, (8, 7, 7), (4, 3, 3))  # noqa: E731
elif which == "msda":
    shapes = torch.tensor([[45, 80], [23, 40], [12, 20], [6, 10]])
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = 4820
    value = torch.randn(8, S, 8, 32, generator=g).to(dev)
    loc = (torch.rand(8, S, 8, 4, 4, 2, generator=g) * 0.1 + torch.rand(8, S, 1, 1, 1, 2, generator=g) * 0.9).to(dev)
    w = torch.softmax(torch.randn(8, S, 8, 16, generator=g), -1).view(8, S, 8, 4, 4).to(dev)
    shapes, lsi = shapes.to(dev), lsi.to(dev)
    fn = lambda: hot_ops.msda_forward(value, shapes, lsi, loc, w)  # noqa: E731
elif which == "vlf":
    q = torch.randn(28800, 1, 256, generator=g).to(dev)
    k = torch.randn(10, 1, 256, generator=g).to(dev)
    v = torch.randn(10, 1, 256, generator=g).to(dev)
    fn = lambda: hot_ops.mha_core(q, k, v, 8)  # noqa: E731
elif which.startswith("k20"):
    # K20 at three of its call sites: encoder FFN up-projection (+ReLU), stage-1 qkv with the LayerNorm in front,
    # stage-2 fc1 + GELU
    M, N, K, act, use_ln = {"k20ffn": (38560, 2048, 256, "relu", False), "k20qkv1": (28800, 576, 192, "none", True),
                            "k20fc1s2": (7360, 1536, 384, "gelu", False)}[which]
    x = torch.randn(M, K, generator=g).to(dev)
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ln = ((torch.rand(K, generator=g) + 0.5).to(dev), torch.randn(K, generator=g).to(dev), 1e-5) if use_ln else None
    stats = hot_ops.row_stats(x, 1e-5) if use_ln else None
    fn = lambda: hot_ops.linear_split(x, wt, b, ln=ln, act=act, stats=stats)  # noqa: E731
elif which.startswith("k13"):
    # K13b at three of its call sites: stage-0 qkv (LayerNorm in front), stage-0 fc1 + GELU, stage-2 qkv (step-by-step form)
    M, N, K, act, use_ln = {"k13qkv0": (115200, 288, 96, "none", True), "k13fc1s0": (115200, 384, 96, "gelu", True),
                            "k13qkv2": (7360, 1152, 384, "none", False)}[which]
    x = torch.randn(M, K, generator=g).to(dev)
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ln = ((torch.rand(K, generator=g) + 0.5).to(dev), torch.randn(K, generator=g).to(dev), 1e-5) if use_ln else None
    fn = lambda: hot_ops.ws_linear(x, wt, b, ln, None, act)  # noqa: E731
elif which.startswith("k23"):
    # K23 at its four call sites: Video-Swin stage 0 / 1 / 2 MLP (norm2 + fc1 + GELU + fc2 + shortcut), the encoder's FFN + norm2
    M, Cw, F, act, use_ln = {"k23s0": (115200, 96, 384, "gelu", True), "k23s1": (28800, 192, 768, "gelu", True),
                             "k23s2": (7360, 384, 1536, "gelu", True), "k23enc": (38560, 256, 2048, "relu", False)}[which]
    M *= CLIPS
    x = torch.randn(M, Cw, generator=g).to(dev)
    w1, b1 = (torch.randn(F, Cw, generator=g) / Cw ** 0.5).to(dev), torch.randn(F, generator=g).to(dev)
    w2, b2 = (torch.randn(Cw, F, generator=g) / F ** 0.5).to(dev), torch.randn(Cw, generator=g).to(dev)
    ln = ((torch.rand(Cw, generator=g) + 0.5).to(dev), torch.randn(Cw, generator=g).to(dev), 1e-5)
    fn = lambda: hot_ops.mlp_split(x, w1, b1, w2, b2, act, ln if use_ln else None, x, post_ln=None if use_ln else ln)  # noqa: E731
elif which.startswith("k24"):
    # K24 at four of its call sites: stage-2 qkv, the encoder's value_proj, stage-3 norm1 + qkv, the fusion query projection
    M, N, K, use_ln = {"k24qkv2": (7360, 1152, 384, False), "k24enc": (38560, 256, 256, False),
                       "k24qkv3": (1920, 2304, 768, True), "k24vlf": (28800, 256, 256, False)}[which]
    M *= CLIPS
    x = torch.randn(M, K, generator=g).to(dev)
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ln = ((torch.rand(K, generator=g) + 0.5).to(dev), torch.randn(K, generator=g).to(dev), 1e-5) if use_ln else None
    fn = lambda: hot_ops.xs_linear(x, wt, b, ln, None, "none")  # noqa: E731
elif which.startswith("ln"):
    rows, C = {"ln0": (115200, 96), "ln1": (28800, 192), "lnenc": (38560, 256), "ln2": (7360, 384)}[which]
    x = torch.randn(rows, C, generator=g).to(dev)
    y = torch.randn(rows, C, generator=g).to(dev)
    w, b = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    fn = lambda: hot_ops.add_layernorm(x, y, w, b)  # noqa: E731
else:
    feats = torch.randn(8, 8, 90, 160, generator=g).to(dev)
    params = torch.randn(160, 169, generator=g).to(dev)
    refs = torch.rand(160, 2, generator=g).to(dev)
    fn = lambda: hot_ops.dynamic_mask(feats, params, refs, (360, 640))  # noqa: E731
for _ in range(3):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    fn()
e.record()
torch.cuda.synchronize()
print(which, "avg us", 1e3 * s.elapsed_time(e) / reps)
