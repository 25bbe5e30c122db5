"""Summarize one steady-state forward from a rocprofv3 --kernel-trace CSV.
usage: python tools/analyze_trace.py <kernel_trace.csv> [--top N]"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spin_kernel" not in r["Kernel_Name"]]  # drop torch.cuda._sleep
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import os  # noqa: E402
PER = int(os.environ.get("SOC_TRACE_CLIPS_PER_GROUP", "1"))      # pair pipeline: a replay (2 clips) ends with two dyn_mask launches
idx = [i for i, r in enumerate(rows) if "dyn_mask" in r["Kernel_Name"]][PER - 1::PER]
# a steady-state replay: of the segments between two delimiters that hold a whole replay's kernels (head of one group beside
# the tail of the previous one; the passes of bench.py begin with a head-only replay, end with a tail-only one and are
# separated by host work), the one that took the least wall time (delimiter to delimiter)
_segs = [(idx[j] - idx[j - 1], int(rows[idx[j]]["End_Timestamp"]) - int(rows[idx[j - 1]]["End_Timestamp"]), j) for j in range(1, len(idx))]
if "--segments" in sys.argv:            # every segment between two delimiters: kernels, wall ms
    for n, w, j in _segs:
        print(f"segment {j:3d}: {n:5d} kernels  {w / 1e6:8.2f} ms")
    sys.exit(0)
_med = sorted(n for n, _, _ in _segs)[len(_segs) // 2]
k = min((w, j) for n, w, j in _segs if n >= 0.9 * _med)[1]
seg = rows[idx[k - 1] + 1: idx[k] + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])


def cat(n):
    if any(k in n for k in ("win_attn", "msda", "xattn", "dyn_mask", "add_layernorm", "linear_small", "box_refine",
                            "upsample_threshold", "upsample_merge", "upsample_add_nchw", "resize_", "gemm_nt_kernel", "gn_stats", "gn_apply",
                            "patch_merge", "ws_linear", "dec_cross_attn", "row_mlp", "groupnorm_nchw", "conv3x3_tokens",
                            "linear_split_kernel", "row_stats_kernel", "patch_embed", "split_pack_kernel", 
                            "mlp_split_kernel", "mlp_reduce_kernel", "mlp_pack_kernel", "xs_linear_kernel", "xs_pack_kernel", "small_attn_kernel")):
        return "soc_hip kernels"
    if n.startswith("Cijk"):
        return "GEMM (hipBLASLt/rocBLAS)"
    if "layer_norm" in n:
        return "layer_norm"
    if "conv" in n.lower() or "Im2d2Col" in n or "Sp3Asm" in n or "igemm" in n.lower():
        return "conv (MIOpen)"
    if "direct_copy" in n or "copy" in n.lower():
        return "copy/contiguous"
    if "CatArray" in n:
        return "cat"
    if "Gelu" in n:
        return "gelu"
    if "elementwise" in n:
        return "elementwise other"
    return "other"


agg = collections.defaultdict(lambda: [0, 0])
names = collections.defaultdict(collections.Counter)
big = []
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    c = cat(r["Kernel_Name"])
    agg[c][0] += d
    agg[c][1] += 1
    names[c][r["Kernel_Name"][:110]] += d
    big.append((d, r["Kernel_Name"][:90], r["Grid_Size_X"]))
busy = sum(v[0] for v in agg.values())
print(f"forward {k} of the trace: {len(seg)} kernels, wall {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms")
for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{d / 1e6:8.3f} ms {n:5d}  {k}")
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
if top:
    for c in ("elementwise other", "copy/contiguous", "layer_norm", "other"):
        print("---", c)
        for n, d in names[c].most_common(6):
            print(f"   {d / 1e6:7.3f} ms  {n}")
    print("--- largest single launches")
    for d, n, gx in sorted(big, reverse=True)[:top]:
        print(f"   {d / 1e3:8.1f} us grid {gx:>9s}  {n}")
