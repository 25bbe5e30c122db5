"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected SEPARATELY as MI355X_MICROARCH.md's
HBM section prescribes) of `bench.py --eager --steps G --warmup G --no-cpu-baseline` (G = clips per launch group) into HBM bytes per clip for
every hand-written kernel.  usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_...csv> > json
Counter unit = KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads, so it is doubled
(all these kernels read 16 B per lane); WRITE_SIZE is exact for 16-B stores."""
import collections
import csv
import json
import sys

GROUPS = collections.OrderedDict([
    ("win_attn3d", ("win_attn3d",)), ("linear_split", ("::linear_split_kernel",)), ("row_stats", ("row_stats_kernel",)), ("mlp_split", ("mlp_split_kernel", "mlp_reduce_kernel")),
    ("xs_linear", ("xs_linear_kernel",)),
    ("msda_fwd", ("msda_fwd", "msda_fused")), ("xattn", ("xattn_",)), ("dyn_mask", ("dyn_mask",)),
    ("add_layernorm", ("add_layernorm",)), ("linear_small", ("linear_small",)), ("linear_act", ("gemm_nt_kernel",)),
    ("groupnorm_tokens", ("gn_stats", "gn_apply")), ("patch_merge_layernorm", ("patch_merge",)), ("patch_embed_layernorm", ("patch_embed_ln",)), ("ws_linear", ("ws_linear_kernel", "ws_linear_split_kernel")),
    ("box_refine", ("box_refine",)), ("decoder_cross_attn", ("dec_cross_attn_kernel",)), ("row_mlp", ("row_mlp_kernel",)),
    ("fpn_elementwise", ("groupnorm_nchw_kernel", "upsample_add_nchw_kernel", "upsample_add_tokens_kernel")),
    ("conv3x3_tokens", ("conv3x3_tokens_kernel",)),
])


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # forwards are delimited by the single dyn_mask launch each one ends with
    ends = [i for i, r in enumerate(rows) if "dyn_mask" in r["Kernel_Name"]]
    return rows, ends


PER = int(sys.argv[3]) if len(sys.argv) > 3 else 1       # clips per launch group (bench.py --pipeline pairs: 2): a group ends with PER dyn_mask launches


def per_forward(rows, ends):
    """sum the counter per kernel group over one steady-state launch group (PER clips: one head, PER tails), per CLIP: the
    second timed group (group 0 is the warm-up; behind the timed steps bench.py runs its instrumented pass and the
    back-to-back family replays, which are not forwards)"""
    i = 2 if len(ends) > 2 * PER else len(ends) // PER - 1
    seg = rows[ends[i * PER - 1] + 1: ends[(i + 1) * PER - 1] + 1]
    out = collections.defaultdict(lambda: [0.0, 0])
    for r in seg:
        for g, keys in GROUPS.items():
            if any(k in r["Kernel_Name"] for k in keys):
                out[g][0] += float(r["Counter_Value"]) * 1024.0 / PER
                out[g][1] += 1
    return out


fetch = per_forward(*load(sys.argv[1], "FETCH_SIZE"))
write = per_forward(*load(sys.argv[2], "WRITE_SIZE"))
res = {"clips_per_launch_group": PER,
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate runs) -- python3 bench.py --eager "
                  "--steps G --warmup G --no-cpu-baseline --no-stream --no-f32-pass (G = clips_per_launch_group; tools/pmc_run_r05.sh); "
                  "tools/pmc_traffic.py",
       "note": "counter unit = KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the bytes of "
               "wide (16 B/lane) coalesced reads, so fetch bytes are doubled; WRITE_SIZE is exact for 16-B stores. "
               "Second timed forward of the run (steady state).",
       "per_clip_bytes": {}}
for g in GROUPS:
    if g in fetch or g in write:
        f, n = fetch.get(g, [0.0, 0])
        w, _ = write.get(g, [0.0, 0])
        res["per_clip_bytes"][g] = {"launches_per_group": n, "fetch_raw": f, "fetch_corrected": 2 * f, "write": w,
                                   "hbm_total": 2 * f + w}
print(json.dumps(res, indent=1))
