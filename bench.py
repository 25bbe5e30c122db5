#!/usr/bin/env python
"""Headline benchmark: clips/s of SOC's per-clip inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without an outer launcher: this process starts N ranks itself (one per GPU, before it touches the GPU,
clip_parallel.spawn_ranks = the reference's mp.Process fan-out, infer_refytb.py:84-109) and only waits; under
`python -m torch.distributed.run --nproc-per-node N` the ranks already exist and WORLD_SIZE must equal --gpus.

A step = one eval forward of Video-Swin-T SOC on one synthetic clip [T=8,3,360,640] (random
deterministic weights, pre-tokenised 10-token expression) + query selection, i.e. the body of
the reference's inference loop (infer_refytb.py:206-227).  Since round 5 a group of independent clips shares each launch of the
forward (graph_runner.group_pipeline_class: DEFAULT_GROUP = 10 clips, fixed since round 6); the VOC module -- the one place where the
reference's forward couples the clips of a batch -- runs per clip, so every clip gets its single-clip (B = 1) result, and EVERY
timed record is checked against the same clip's one-clip-per-launch record (every slot of a group holds a different clip);
`single_clip_ms_per_step` is the one-clip-per-launch pipeline of rounds 1-4 beside it.  Inputs are resident in HBM before the
timed region.  Clips shard over ranks (weak scaling: K clips per rank, no data-path collective);
the single result all_gather sits inside the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec
HBM_ACHIEVABLE_GBS = 6300.0    # MI355X_MICROARCH.md: measured float4 copy rate (79 % of spec): what a streaming kernel can reach
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak (spec; the clock held on random data is lower)
_REL_LATE = os.environ.get('BENCH_REL_LATE', '0') == '1'
_NO_RECORD_COPY = os.environ.get('BENCH_NO_RECORD_COPY', '0') == '1'
_STREAM_MODE = os.environ.get('BENCH_STREAM_MODE', 'full')     # diagnostics of the streamed pass: nowait | d2d
_FEED_DEPTH = int(os.environ.get('BENCH_FEED_DEPTH', '2'))     # device slots of the streamed pass's feeder
WEIGHT_SEED = 2023
PROFILE_TAG = "r06"            # the round whose rocprofv3 summaries under profiles/ belong to this bench.py
MAX_LINE_BYTES = 8000          # the driver reads the line from a bounded stdout tail: round 4's 21.5 KB line did not parse
FLIP_WINDOW = 6e-5             # |reference logit| below which a thresholded pixel may differ: the reference's own
                               # 1-vs-8-thread noise at this logit scale; everywhere else masks are bit-exact


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backbone", default="video-swin-t")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one clip per graph replay (ClipGraph) instead of the software-pipelined PipelinedClipGraph")
    ap.add_argument("--pipeline", default=None,
                    type=lambda v: v if v in ("two-stream", "one-graph", "pairs", "quads", "octs") or (v.startswith("group") and v[5:].isdigit())
                    else (_ for _ in ()).throw(argparse.ArgumentTypeError("two-stream | one-graph | pairs | quads | octs | group<N>")),
                    help="software pipeline across clips (default: default_pipeline()): group<N> / octs / quads / pairs = N / 8 / 4 / 2 "
                         "independent clips per launch group, VOC over independent clips (graph_runner.group_pipeline_class); "
                         "one-graph = one clip per launch (rounds 1-4, PipelinedClipGraph; always timed beside the headline as "
                         "single_clip_ms_per_step); two-stream")
    ap.add_argument("--no-stream", action="store_true",
                    help="skip the second, H2D-inclusive timed pass (stream_ms_per_step)")
    ap.add_argument("--stub", action="store_true",
                    help="CPU / gloo dry run of the rank loop with a stub clip step (no model, no kernels): what the 2-rank "
                         "CPU test drives; the line it prints carries \"stub\": true and is not a measurement")
    ap.add_argument("--all-passes", action="store_true",
                    help="N > 1: also run the streamed and the f32-only passes (by default a rank of a multi-GPU run does the "
                         "headline pass only: one graph capture per rank)")
    ap.add_argument("--detail", default=os.environ.get("BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where rank 0 writes the long form (per-family / per-shape rooflines, per-kernel times, every parity "
                         "number, the CPU runs); the ONE JSON line on stdout stays under MAX_LINE_BYTES")
    ap.add_argument("--no-single-pass", action="store_true",
                    help="pair pipeline: skip the extra timed pass with one clip per head launch (single_clip_ms_per_step)")
    ap.add_argument("--no-f32-pass", action="store_true",
                    help="skip the extra timed pass with every GEMM on the f32 MFMA path (SOC_MATMUL=f32 arithmetic)")
    return ap.parse_args()


def granted_cpus(cgroup_root="/sys/fs/cgroup"):
    """CPUs this process may actually use: the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us) rounded
    up, capped by the affinity mask.  The GPU boxes show 256 logical CPUs behind a 16-CPU quota: timing the CPU
    baseline at 128 'physical' threads there measures oversubscription, not the machine."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open(os.path.join(cgroup_root, "cpu.max")) as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = -(-int(q) // int(per))
    except (OSError, ValueError):
        try:
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")) as f:
                q = int(f.read())
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")) as f:
                per = int(f.read())
            if q > 0:
                quota = -(-q // per)
        except (OSError, ValueError):
            pass
    return max(1, min(avail, quota) if quota else avail), quota, avail


def _cpu_topology():
    """(physical cores this process may run on, CPU model string) from /proc/cpuinfo."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    cores, model, cpu = set(), "unknown", None
    phys_id = core_id = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f.read().split("\n") + [""]:
                if ln.startswith("processor"):
                    cpu = int(ln.split(":")[1])
                elif ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    phys_id = int(ln.split(":")[1])
                elif ln.startswith("core id"):
                    core_id = int(ln.split(":")[1])
                elif not ln.strip() and cpu is not None:
                    if cpu in allowed:
                        cores.add((phys_id, core_id if core_id is not None else cpu))
                    cpu = phys_id = core_id = None
    except OSError:
        pass
    return max(len(cores), 1), model


GOLDENS = {("video-swin-t", 8, 360, 640): ("full_forward.npz", 1), ("video-swin-b", 8, 360, 640): ("full_forward_b.npz", 1),
           ("video-swin-b", 8, 720, 1280): ("full_forward_b720.npz", 3)}


def golden_cfg(backbone, T, H, W):
    """(file under tests/golden/ or None, seed of its clip and token ids) of the reference-generated golden of a configuration."""
    return GOLDENS.get((backbone, T, H, W), (None, 1))


def headline(a, world, timed, workload, launch):
    """The contract's keys of the JSON line, from what the measured rank loop returned (shared by the stub dry run)."""
    dt = timed["seconds"]
    T, H, Wd = a.frames, a.height, a.width
    return {
        "metric": "clips/s (T=8, 360x640, Video-Swin-T)" if (a.backbone, T, H, Wd) == ("video-swin-t", 8, 360, 640)
                  else f"clips/s (T={T}, {H}x{Wd}, {a.backbone})",
        "value": world * a.steps / dt, "unit": "clips/s", "n_gpus": world, "ranks_seen": timed["ranks_seen"],
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "seconds_per_rank": timed.get("seconds_per_rank"),
        "config": {"workload": workload, "clips_per_rank": a.steps,
                   "parallelism": f"clip-parallel x{world}, one result all_gather", "launch": launch}}


_ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "peak_basis", "vs_f32_mfma_peak", "unit", "frac", "frac_of_ceiling", "traffic", "launches", "clips_per_replay",
                  "algorithmic_flop_per_clip", "algorithmic_bytes_per_clip", "avg_launch_us", "ms_per_clip", "source")


DEFAULT_GROUP = 10             # clips per launch group up to 360 x 640 (same box, Swin-T: 4.81-4.85 ms per clip in eights / tens / twelves)


def default_pipeline(steps: int, frames: int, height: int, width: int) -> str:
    """The pipeline bench.py times when --pipeline is not given.  Up to 360x640 several independent clips share every launch
    (graph_runner.group_pipeline_class): per clip 5.13 ms in fours, 4.81-4.84 in eights / tens / twelves, 4.89 in sixteens (same
    box, Swin-T).  The group is FIXED -- DEFAULT_GROUP clips, whatever --steps is (round 5 took the first of 8, 10, 12, ... that
    divided --steps, i.e. a group tuned to the driver's step count): a clip count that is not a multiple runs a part-filled last
    group at the cost of a whole replay, inside the timed region.  Fewer clips than a group: one group of all of them.  Above
    360x640 a clip nearly fills the chip by itself: pairs (Swin-B 720p: 42.1 ms per clip in pairs, 42.6 alone, 49.2 in fours)."""
    if steps <= 1:
        return "one-graph"
    if frames * height * width > 8 * 360 * 640:
        return "pairs"
    g = min(DEFAULT_GROUP, steps)
    return {8: "octs", 4: "quads", 2: "pairs"}.get(g, f"group{g}")


def compact_line(full):
    """The ONE line the driver parses, cut from the long form: the contract's keys, ONE `roofline` object (the kernel family
    with the largest share of a clip), `cpu_baseline`, a short `parity` block, and one row per other kernel family.  What is
    left out (per-shape tables, per-kernel times, prose, the CPU runs) goes to the --detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "seconds_per_rank", "config", "matmul", "stub", "records_ok",
            "f32_mfma_only_ms_per_step", "stream_ms_per_step", "single_clip_ms_per_step", "clips_per_head_launch",
            "long_run_ms_per_step", "long_run_steps", "kernels_per_forward", "switches")
    line = {k: full[k] for k in keep if k in full}
    if "roofline" in full:
        line["roofline"] = {k: full["roofline"][k] for k in _ROOFLINE_KEYS if k in full["roofline"]}
        fams = {}
        for k, v in full.items():
            if k.startswith("roofline_") and k != "roofline_other":
                fams[k[len("roofline_"):]] = {"bound": v["bound"], "frac": round(v["frac"], 4),
                                              "frac_of_ceiling": round(v["frac_of_ceiling"], 4),
                                              "ms_per_clip": round(v["ms_per_clip"], 4), "launches": v.get("launches"),
                                              "traffic_over_algorithmic": (round(v["traffic"] / v["algorithmic_bytes_per_clip"], 3)
                                                                           if v.get("traffic") else None)}
        for k, v in full.get("roofline_other", {}).items():
            fams[k] = {"bound": "hbm", "frac": round(v["frac"], 4), "ms_per_clip": round(v["ms_per_clip"], 4),
                       "avg_launch_us": round(v["avg_launch_us"], 1)}
            if "ta_busy_share" in v:          # K2: bound by the CUs' vector-memory path, not by HBM
                fams[k]["texture_path_busy"] = v["ta_busy_share"]
        line["roofline_families"] = fams
    if "cpu_baseline" in full:
        line["cpu_baseline"] = {k: full["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "cpu_model", "sample")}
    p = full.get("parity")
    if p:
        recs = [{"flips": p["timed_path_thresholded_mask_flips"], "at": p["timed_path_max_abs_ref_logit_at_flips"],
                 "d": p["timed_path_mask_logit_max_abs_diff"]}]
        recs += [{"flips": e["thresholded_mask_flips"], "at": e["max_abs_oracle_logit_at_flips"], "d": e["mask_logit_max_abs_diff"]}
                 for e in p.get("timed_path_other_records_vs_cpu_oracle", [])]
        sc = full.get("slot_check")
        line["parity"] = {
            "checked": "records of the timed region: 0 vs the reference golden, 1..3 vs the CPU oracle"
                       + (f"; ALL {sc['records']} (distinct clips, one per group slot) vs the same clip's one-clip-per-launch record"
                          if sc else ""),
            "records": sc["records"] if sc else len(recs),
            "records_vs_golden_or_oracle": len(recs),
            **({"all_records_max_abs_diff_vs_single_clip": sc["max_abs_diff"], "all_records_selected_query_equal": sc["selected_query_equal"],
                "distinct_clips": sc["distinct_clips"]} if sc else {}),
            "selected_query_matches": bool(p["timed_path_selected_query"] == p["timed_path_selected_query_ref"] and all(
                e["selected_query"] == e["selected_query_oracle"] for e in p.get("timed_path_other_records_vs_cpu_oracle", []))),
            "mask_logit_max_abs_diff": max(r["d"] for r in recs),
            "mask_logit_max_abs_diff_vs_cpu_oracle": p.get("timed_path_mask_logit_max_abs_diff_vs_cpu_oracle"),
            "max_abs_logit": p.get("max_abs_logit"),
            "flips_total": sum(r["flips"] for r in recs),
            "max_abs_ref_logit_at_flips": max(r["at"] for r in recs),
            "pixels_per_record": p["timed_path_pixels"],
            "flip_window": FLIP_WINDOW}
    return line


def emit(line, full, detail_path):
    """Write the long form beside the bench, then print the compact line -- the LAST thing on stdout, bounded."""
    if full is not None and detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(full, f, indent=1)
            line["detail"] = os.path.relpath(detail_path, ROOT)
        except OSError as e:          # a read-only tree must not cost the run its line
            line["detail"] = f"not written: {e}"
    text = json.dumps(line)
    # a finished multi-minute run must not end in an AssertionError over a few bytes: optional blocks go (they are in the detail
    # file) until the line fits, and the line says so
    for drop in ("switches", "kernels_per_forward", "roofline_families", "seconds_per_rank", "matmul"):
        if len(text) < MAX_LINE_BYTES:
            break
        if drop in line:
            del line[drop]
            line["truncated"] = line.get("truncated", []) + [drop]
            text = json.dumps(line)
    if len(text) >= MAX_LINE_BYTES and "config" in line:
        line["config"] = {k: (v[:200] if isinstance(v, str) else v) for k, v in line["config"].items()}
        line["truncated"] = line.get("truncated", []) + ["config"]
        text = json.dumps(line)
    sys.stderr.flush()
    print(text, flush=True)


def stub_main(a, CP):
    """`--stub`: the rank loop of this file -- warm-up, the RCCL / gloo warm-up gather, timed_sharded_run, the world and
    ranks_seen check, the JSON line -- on CPU tensors with a stub step (record i of rank r = f(r, i)).  No model, no GPU."""
    rank, local_rank, world = CP.init_rank("cpu", expect_world=a.gpus)
    CP.pin_rank_cpus(rank, world)
    R = 8
    results = torch.zeros(a.steps, R)

    def run_steps(n, out):
        for i in range(n):
            out[i % out.shape[0]] = torch.tensor([float(rank), float(i), float(rank * 1000 + i)] + [1.0] * (R - 3))

    run_steps(a.warmup, results)
    if dist.is_initialized():
        CP.gather_results(results)
    results.zero_()
    timed = CP.timed_sharded_run(lambda out: run_steps(a.steps, out), results, None)
    g = timed["gathered"]
    assert g.shape[0] == world == a.gpus and timed["ranks_seen"] == list(range(world)), timed["ranks_seen"]
    ok = all(g[r, i, 2].item() == r * 1000 + i for r in range(world) for i in range(a.steps))
    if rank == 0:
        line = headline(a, world, timed, "stub step (no model): dry run of the rank loop", "stub")
        line.update(stub=True, records_ok=bool(ok))
        emit(compact_line(line), None, None)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    a = parse()
    if a.pipeline is None:
        a.pipeline = default_pipeline(a.steps, a.frames, a.height, a.width)
    from neurips2023_soc_amd import clip_parallel as CP
    CP.rank_environment()           # before anything touches the GPU: the same process environment in both launch modes
    if a.gpus > 1 and not CP.launched_as_rank():
        # fan out BEFORE any GPU call in this process; the parent never execs, it waits and relays the exit code
        return CP.spawn_ranks(a.gpus, [sys.executable, os.path.abspath(__file__), *sys.argv[1:]])
    if a.stub:
        return stub_main(a, CP)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs an MI355X"
    rank, local_rank, world = CP.init_rank("cuda", expect_world=a.gpus)   # "nccl" is RCCL on ROCm
    CP.pin_rank_cpus(rank, world)   # N ranks share the box's CPU quota: each keeps to its share (captures, launches)
    if world > 1 and not a.all_passes:
        a.no_stream = a.no_f32_pass = a.no_single_pass = True      # one graph capture per rank; the extra passes are single-GPU diagnostics
    dev = torch.device("cuda", local_rank)
    use_dist = dist.is_initialized()

    import neurips2023_soc_amd as S
    from neurips2023_soc_amd import hot_ops, postprocessing as P, weights as W

    model, _, _ = S.build_model(S.default_args(a.backbone, text_encoder_random_init=True))
    sd = W.load_synthetic(model, WEIGHT_SEED)
    model = model.to(dev).eval()

    T, H, Wd, L, Q = a.frames, a.height, a.width, 10, 20
    # Distinct clips: every timed clip its own (seed 1 + 1000 rank + i) up to 32 clips, a pool of 29 beyond that (a prime: over a
    # long run every pool clip then visits every slot of a launch group).  Round 5 cycled FOUR clips, so a ten-clip group held
    # clips 0,1,2,3,0,1,... and a record returned from slot b - 4 instead of slot b could not have been seen.
    n_pool = a.steps if a.steps <= 32 else 29
    clips_cpu = [W.synthetic_clip(1 + 1000 * rank + i, T, H, Wd) for i in range(n_pool)]
    clips = [c.to(dev) for c in clips_cpu]
    # the expression of the configuration's reference golden (seed 1 for the headline one; the 720p golden was made with seed 3
    # for both its clip and its token ids): record (seed - 1) of the timed region is then the golden's forward
    ids_cpu = W.synthetic_token_ids(golden_cfg(a.backbone, T, H, Wd)[1], L)
    text = {"input_ids": ids_cpu.to(dev), "attention_mask": torch.ones_like(ids_cpu).to(dev)}
    pad = torch.zeros(T, 1, H, Wd, dtype=torch.bool, device=dev)
    targets = [[{"size": (H, Wd)}] for _ in range(T)]
    hm, wm = -(-H // 4), -(-Wd // 4)
    results = torch.zeros(a.steps, CP.record_size(T, Q, hm, wm), device=dev)

    def step(i, record=None, clip=None):  # eager launches, one clip
        samples = S.NestedTensor((clips[i % n_pool] if clip is None else clip)[:, None], pad, unpadded=True)
        out = model(samples, None, text, targets)
        idx, masks = P.select_trajectory(out)
        if record is not None:
            CP.pack_record(record, idx, out["pred_cls"][:, 0, :, 0], masks)
        return out

    def step_group(i0, count, records=None):
        """Eager launches of what ONE replay of the timed pipeline runs: head and tail over `per` clips (pool clips i0, i0 + 1,
        ...), VOC per clip (graph_runner.group_tail); per == 1: step()."""
        if per == 1:
            return step(i0, None if records is None else records[i0])
        group = torch.stack([clips[(i0 + b) % n_pool] for b in range(per)], 1)
        text_g = {k: v.expand(per, -1).contiguous() for k, v in text.items()}
        sb = model.forward_head(S.NestedTensor(group, pad.expand(-1, per, -1, -1), unpadded=True), None, text_g)
        from neurips2023_soc_amd.graph_runner import group_tail
        group_tail(model, sb, targets, True, group_records)
        if records is not None:
            records[i0:i0 + count].copy_(group_records[:count])

    graph = None
    pipelined = False
    if not a.eager:
        from neurips2023_soc_amd.graph_runner import ClipGraph, pipeline_class
        # a failed capture fails the run: the headline is the graph replay, never a silent eager timing
        if not a.no_pipeline:
            Pipeline = pipeline_class(a.pipeline)
            graph = Pipeline(model, T, H, Wd, L, dev)             # tail of clip i beside the head of clip i+1
            pipelined = True
        else:
            graph = ClipGraph(model, T, H, Wd, L, dev)            # one capture, replayed per clip
    # clips per replay of the timed pipeline (--eager launches the same groups without a graph)
    per = getattr(graph, "CLIPS", 1) if graph is not None else (
        1 if a.no_pipeline else __import__("neurips2023_soc_amd.graph_runner", fromlist=["x"]).pipeline_class(a.pipeline).CLIPS)
    clips_per_group = per
    group_records = torch.zeros(max(per, 1), results.shape[1], device=dev)      # scratch of the eager group step

    def run_steps(n, out, feed=None):
        """n clips through the chosen path, results into out[i % len(out)].  `feed` = (feeder, host clips): every clip
        then crosses PCIe inside the loop (pinned host buffer -> one of two device slots on a copy stream, one clip
        ahead of the compute stream) as in the reference's loop (infer_refytb.py:206-212); without it the clips are
        the HBM-resident pool."""
        m = out.shape[0]
        feeder, host = feed if feed is not None else (None, None)
        per = getattr(graph, "CLIPS", 1) if graph is not None else clips_per_group      # clips per replay of THIS pass's pipeline
        # copies run `ahead` clips in front of the compute stream: one clip, or -- with launch groups -- a whole group, so that
        # the next group crosses PCIe while this one computes (a slot is free again once its clip has been staged into the
        # graph's static input, and that staging is queued behind the running replay)
        ahead = min(per, feeder.depth - 1) if feeder is not None else 1
        submitted = [0]

        def next_clip(i):
            if feeder is None:
                return clips[i % n_pool]
            while submitted[0] < min(n, i + 1 + ahead):
                feeder.submit(host[submitted[0] % len(host)])
                submitted[0] += 1
            if _STREAM_MODE == "nowait":        # diagnostic: the copies run, the compute stream neither waits for them nor reads them
                feeder._ready[feeder._acquired % feeder.depth] = None
                feeder.acquire()
                return clips[i % n_pool]
            return feeder.acquire()

        done = 0
        in_flight = []                                                    # clips carried by each replay whose records are still due
        for i in range(n):
            clip = next_clip(i)
            if graph is None:
                if per == 1:
                    step(i, out[i % m], clip)
                    done += 1
                elif i % per == per - 1 or i == n - 1:                    # eager: the pool clips of the group, head once
                    step_group(i - i % per, i % per + 1, out)
                    done += i % per + 1
            else:
                if per == 1:
                    graph.stage_inputs(clip, text["input_ids"])
                else:
                    graph.stage_inputs(clip, text["input_ids"], slot=i % per)
                if feeder is not None and not _REL_LATE:
                    feeder.release()       # the slot has been copied into the graph's static input: reusable from here
                if per > 1 and i % per != per - 1 and i != n - 1:
                    continue               # the replay waits for its second clip (an odd last clip goes alone: slot 1 is stale
                                           # and its record is dropped below)
                rec = graph.replay()
                in_flight.append(i % per + 1)
                if feeder is not None and _REL_LATE:
                    feeder.release()
                if not pipelined:
                    out[i % m].copy_(graph.record, non_blocking=True)
                    in_flight.pop()
                    done += 1
                elif rec is not None:      # software pipeline: a replay returns the record(s) of an earlier replay
                    for r in (rec[:in_flight.pop(0)] if per > 1 else [in_flight.pop(0) and graph.record]):
                        if not _NO_RECORD_COPY:
                            out[done % m].copy_(r, non_blocking=True)
                        done += 1
            if feeder is not None and graph is None:
                feeder.release()
        if pipelined:
            for rec in graph.flush():
                for r in (rec[:in_flight.pop(0)] if per > 1 else [rec]):
                    out[done % m].copy_(r, non_blocking=True)
                    done += 1
        assert done == n, (done, n)

    run_steps(a.warmup, results)
    if use_dist:
        CP.gather_results(results)  # RCCL warm-up, outside the timed region
    torch.cuda.synchronize()
    results.zero_()                 # what is checked below can only have been written by the timed region

    if graph is None:
        hot_ops.profile_begin()
    # barrier + sync | exactly K clips (pipeline drained) + the one result all_gather | barrier + sync; max over ranks
    timed = CP.timed_sharded_run(lambda out: run_steps(a.steps, out), results, dev)
    gathered, dt = timed["gathered"], timed["seconds"]
    timed_records = results[:min(a.steps, n_pool)].cpu()      # clip i of the pool <-> record i
    all_timed_records = results.cpu()                         # record i <-> pool clip i % n_pool: every one is checked below

    # Second timed pass, H2D-inclusive (SURVEY 8d config 5 "stream with per-clip seeds seed0 + i", 8e "pinned,
    # double-buffered H2D"): the same loop, but every clip is copied from a pinned host buffer inside the timed region.
    # `value` stays the resident number; this one is reported beside it as stream_ms_per_step.
    stream = None
    if not a.no_stream:
        from neurips2023_soc_amd.clip_io import DoubleBufferedH2D
        n_host = n_pool                                  # the pool's clips as pinned host buffers (22 MB each), cycled like the pool
        host = [h.pin_memory() for h in clips_cpu]
        if _STREAM_MODE == "d2d":               # diagnostic: the same feeder and events, device-resident sources (no PCIe)
            host = [h.to(dev) for h in host]
        feeder = DoubleBufferedH2D((T, 3, H, Wd), torch.float32, dev, depth=max(_FEED_DEPTH, clips_per_group + 1))
        # A driver recycles a few pinned buffers (clip_io.PinnedPool), so every buffer it copies from has been through the
        # DMA engine before; the first transfer out of a fresh pinned allocation is several times slower than the 0.41 ms
        # (54 GB/s) of the later ones.  Each host clip is therefore copied once, untimed, before the pass.
        warm = torch.empty((T, 3, H, Wd), dtype=torch.float32, device=dev)
        for h in host:
            warm.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        del warm
        sres = torch.zeros_like(results)
        run_steps(max(a.warmup, 2), sres, (feeder, host))
        torch.cuda.synchronize()
        sres.zero_()
        st = CP.timed_sharded_run(lambda out: run_steps(a.steps, out, (feeder, host)), sres, dev)
        stream = {"seconds": st["seconds"], "records": sres[:min(a.steps, n_pool)].cpu(), "n_host": n_host,
                  "max_abs_diff_all_records": float((sres.cpu() - all_timed_records).abs().max())}
        del host, feeder
    if graph is None:
        prof = hot_ops.profile_end()
    else:
        # HIP events cannot bracket nodes inside a graph replay, so the per-kernel durations for the
        # roofline come from an instrumented eager pass over the same clips right after the timed
        # region (same kernels, same inputs, same stream).
        hot_ops.profile_begin()
        for i in range(0, a.steps, per):
            # give the host a head start so the launches queue back to back: an event pair then
            # brackets the kernel alone, not the Python time between record() and launch
            torch.cuda._sleep(60_000_000)
            step_group(i, min(per, a.steps - i), results)
        prof = hot_ops.profile_end()
    prof_clips = -(-a.steps // per) * per      # clips the instrumented pass ran (whole groups)

    # Third timed pass: the same loop with the pixel-sized linear layers on the f32 MFMA path (K13 / K12 / library) instead
    # of K20's three-way bf16 split -- the round-2 arithmetic, reported beside the headline so both are on record.
    f32_pass = None
    if graph is not None and not a.no_f32_pass and hot_ops.split_enabled():
        from neurips2023_soc_amd.graph_runner import ClipGraph
        model.matmul_mode = "f32"        # a property of the model (thread-local inside its forward, a launch argument below)
        try:
            g32 = (Pipeline if pipelined else ClipGraph)(model, T, H, Wd, L, dev)
            main_graph, graph = graph, g32
            r32 = torch.zeros_like(results)
            run_steps(min(a.warmup, 2), r32)
            torch.cuda.synchronize()
            t32 = CP.timed_sharded_run(lambda out: run_steps(a.steps, out), r32, dev)
            f32_pass = {"seconds": t32["seconds"], "record0": r32[0].cpu()}
            graph = main_graph
            del g32
        finally:
            model.matmul_mode = None

    # Fourth timed pass (group pipelines only): the one-clip-per-launch pipeline of rounds 1-4 on the same box, so that the line
    # carries both numbers.
    single_pass = None
    if graph is not None and pipelined and per > 1 and not a.no_single_pass and world == 1:
        from neurips2023_soc_amd.graph_runner import PipelinedClipGraph
        main_graph, graph = graph, PipelinedClipGraph(model, T, H, Wd, L, dev)
        r1 = torch.zeros_like(results)
        run_steps(min(a.warmup, 2), r1)
        torch.cuda.synchronize()
        t1 = CP.timed_sharded_run(lambda out: run_steps(a.steps, out), r1, dev)
        single_pass = {"seconds": t1["seconds"], "records": r1.cpu(),
                       "max_abs_diff": float((r1.cpu() - all_timed_records).abs().max())}
        graph = main_graph

    # Long run (short --steps only): the timed region of the driver's command is two replays; the same loop over ten launch
    # groups is the steadier figure, reported beside the headline (never instead of it).
    long_run = None
    if graph is not None and pipelined and per > 1 and a.steps < 40 and world == 1 and not a.no_single_pass:
        n_long = 10 * per
        rl = torch.zeros(n_long, results.shape[1], device=dev)
        tl = CP.timed_sharded_run(lambda out: run_steps(n_long, out), rl, dev)
        long_run = {"seconds": tl["seconds"], "steps": n_long}
        del rl

    # Every record of the timed region against the SAME clip through the one-clip-per-launch path (the single-clip pass above when
    # it ran, eager single-clip forwards otherwise): whichever slot of whichever group a clip sat in, it must come back with its
    # own B = 1 result (selected query equal, record within 1e-4).  Records 0..3 are additionally checked against the reference
    # golden / the CPU oracle further down.
    slot_check = None
    if per > 1 and world == 1 and graph is not None:         # (an --eager run has no graph slots to mix up, and the PMC passes count its launches)
        if single_pass is not None:
            ref_recs, how = single_pass["records"], "single-clip pipeline pass (PipelinedClipGraph)"
        else:
            ref_dev = torch.zeros_like(results)
            for i in range(a.steps):
                step(i, ref_dev[i])
            ref_recs, how = ref_dev.cpu(), "eager single-clip forwards"
        d_each = (ref_recs - all_timed_records).abs().amax(1)
        q_same = bool((ref_recs[:, 0] == all_timed_records[:, 0]).all())
        distinct = float((all_timed_records[1:min(a.steps, n_pool)] - all_timed_records[0]).abs().amax(1).min()) if min(a.steps, n_pool) > 1 else None
        slot_check = {"records": int(a.steps), "against": how, "max_abs_diff": float(d_each.max()), "selected_query_equal": q_same,
                      "distinct_clips": min(a.steps, n_pool), "min_abs_diff_between_different_clips": distinct}
        # a mixed-up slot is a difference of order 1 (different clips); the run-to-run noise of a record is 5-6e-5.  The hard failure is
        # at the north_star tolerance -- a finished run must not lose its line to a box with a little more noise -- and the committed
        # line is held to 1e-4 by tests/test_bench_contract.py
        assert q_same and slot_check["max_abs_diff"] < 1e-3, slot_check
        assert distinct is None or distinct > 1e-2, slot_check      # the pool's clips really differ: a swapped slot would show

    # The dominant kernel families: replay the launches of ONE forward back to back between one HIP-event pair on the launch
    # stream (per-launch event pairs add host / queue latency to 30-400 us kernels), with the forward's own tensors.  Every
    # launch carries its algorithmic FLOPs and algorithmic HBM bytes, so that it can be priced against the ceiling that
    # binds IT (matrix cores or HBM) -- see ceiling() below.
    K1_REPS = 20

    def replay_calls(fn, calls, reps, star=False):
        torch.cuda._sleep(40_000_000)        # head start for the host so the launches queue back to back
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
        for _ in range(reps):
            for c in calls:
                fn(*c) if star else fn(**c)
        e_ev.record()
        torch.cuda.synchronize()
        return s_ev.elapsed_time(e_ev) / reps

    def k1_cost(c):
        qkv, _, _, n_heads, window, shift = c
        B_, D_, H_, W_, C3 = qkv.shape
        Cc = C3 // 3
        win, _ = hot_ops.clamp_window((D_, H_, W_), window, shift)
        n_win = B_ * -(-D_ // win[0]) * -(-H_ // win[1]) * -(-W_ // win[2])
        n_tok = win[0] * win[1] * win[2]
        return (4.0 * n_tok * n_tok * (Cc // n_heads) * n_win * n_heads, 4.0 * qkv.numel() * 4 / 3,
                (tuple(qkv.shape), n_heads, tuple(shift) != (0, 0, 0)))

    def linear_cost(c):
        Nn, Kk = c["weight"].shape
        Mm = c["x"].numel() // Kk
        extra = sum(1 for k in ("residual", "mul") if c.get(k) is not None) * Mm * Nn + (Mm * Kk if c.get("add") is not None else 0)
        return (2.0 * Mm * Nn * Kk, 4.0 * (Mm * Kk + Mm * Nn + Nn * Kk + extra),
                (Mm, Nn, Kk, c.get("ln") is not None, c.get("act", "none"), c.get("residual") is not None))

    def mlp_cost(c):
        Ff, Cc = c["w1"].shape
        Mm = c["x"].numel() // Cc
        return (4.0 * Mm * Ff * Cc,
                4.0 * (Mm * Cc * (2 + (c["residual"] is not None) + bool(c.get("return_sum"))) + 2 * Ff * Cc),
                (Mm, Cc, Ff, c["act"], c["ln"] is not None, c["post_ln"] is not None, bool(c.get("return_sum"))))

    families = {}
    for fam, rec, fn, cost, star in (
            ("win_attn3d", hot_ops.record_window_attention_calls, hot_ops.window_attention3d, k1_cost, True),
            ("linear_split", hot_ops.record_linear_split_calls, hot_ops.linear_split, linear_cost, False),
            ("ws_linear", hot_ops.record_ws_linear_calls, hot_ops.ws_linear, linear_cost, False),
            ("xs_linear", hot_ops.record_xs_linear_calls, hot_ops.xs_linear, linear_cost, False),
            ("mlp_split", hot_ops.record_mlp_split_calls, hot_ops.mlp_split, mlp_cost, False)):
        rec(True)
        step_group(0, per)           # the launches of one replay of the timed pipeline: `per` clips
        calls = rec(False)
        torch.cuda.synchronize()
        if not calls:
            continue
        replay_calls(fn, calls, 3, star)
        ms = replay_calls(fn, calls, K1_REPS, star)
        shapes, seen = [], {}
        for c in calls:
            seen.setdefault(cost(c)[2], []).append(c)
        for key, cs in seen.items():
            fl, by, _ = cost(cs[0])
            us = 1e3 * replay_calls(fn, cs[:1], K1_REPS, star)
            shapes.append({"shape": list(key), "launches": len(cs), "us": round(us, 1), "flop": fl, "bytes": by})
        families[fam] = {"ms": ms, "launches": len(calls), "shapes": shapes}      # ms, launches: of ONE replay = `per` clips
        del calls[:]

    assert gathered.shape[0] == world == a.gpus and timed["ranks_seen"] == list(range(world)), timed["ranks_seen"]

    if rank == 0:
        line = {
            **headline(a, world, timed,
                       f"SOC eval forward + query selection per clip (one clip + one expression = the reference's B=1 forward), "
                       f"{a.backbone}, T={T}, {H}x{Wd}, L={L} tokens, random deterministic weights (seed {WEIGHT_SEED})"
                       + (f"; {per} independent clips share each head launch" if per > 1 else ""),
                       "eager" if graph is None else (
                           ("hipGraph replays, software-pipelined on two streams: Video-Swin + fusion + encoder of clip i | text "
                            "encoder of clip i, tail of clip i-1" if Pipeline.__name__ == "TwoStreamClipGraph" else
                            f"hipGraph replay, software-pipelined, {Pipeline.CLIPS} independent clips per launch group: head (Video-Swin, fusion, "
                            "encoder) of group i beside the tail of group i-1, VOC -- the one module of the reference that couples "
                            "a batch -- run per clip" if Pipeline.CLIPS > 1 else
                            "hipGraph replay, software-pipelined: tail of clip i beside the head of clip i+1")
                           if pipelined else "hipGraph replay (one graph per clip geometry)")),
            "matmul": ("f32 in / out / accumulate; large products as 6 bf16 MFMA products of a 3-way operand split (f32-grade: "
                       "error vs f64 <= the f32 library GEMM's, test_linear_split_is_f32_grade); f32_mfma_only_* = no split")
                      if hot_ops.split_enabled() else "f32 MFMA (SOC_MATMUL=f32)",
            **({"f32_mfma_only_ms_per_step": 1e3 * f32_pass["seconds"] / a.steps,
                "f32_mfma_only_value": world * a.steps / f32_pass["seconds"],
                "f32_mfma_only_record0_max_abs_diff": float((f32_pass["record0"] - timed_records[0]).abs().max())}
               if f32_pass is not None else {}),
            **({"single_clip_ms_per_step": 1e3 * single_pass["seconds"] / a.steps,
                "single_clip_value": world * a.steps / single_pass["seconds"],
                "single_clip": "the same loop with ONE clip per head launch (graph_runner.PipelinedClipGraph: the pipeline of rounds "
                               "1-4) on the same box; its records differ from the timed ones by single_clip_max_abs_diff",
                "single_clip_max_abs_diff": single_pass["max_abs_diff"]} if single_pass is not None else {}),
            "clips_per_head_launch": per,
            **({"long_run_ms_per_step": 1e3 * long_run["seconds"] / long_run["steps"], "long_run_steps": long_run["steps"]}
               if long_run is not None else {}),
            **({"slot_check": slot_check} if slot_check is not None else {}),
            **({"stream_ms_per_step": 1e3 * stream["seconds"] / a.steps, "stream_value": world * a.steps / stream["seconds"],
                "stream": f"same loop with every clip copied host->device inside the timed region: {stream['n_host']} "
                          "pinned host clips (seeds seed0 + i; each buffer DMA-ed once before the pass, as a recycled pinned pool is), "
                          "two device slots (a slot is released right behind the copy into the graph's static input), copy stream one clip ahead "
                          "(clip_io.DoubleBufferedH2D); 22 MB per clip at 360x640",
                "stream_record0_max_abs_diff_vs_resident": float((stream["records"][0] - timed_records[0]).abs().max()),
                "stream_all_records_max_abs_diff_vs_resident": stream["max_abs_diff_all_records"]}
               if stream is not None else {}),
        }
        # HBM bytes per clip from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected
        # separately and corrected as MI355X_MICROARCH.md prescribes): profiles/r01_hbm_traffic_pmc.json
        traffic, traffic_file = {}, None
        for name in ("r06_hbm_traffic_pmc.json", "r05_hbm_traffic_pmc.json", "r04_hbm_traffic_pmc.json", "r03_hbm_traffic_pmc.json", "r02_hbm_traffic_pmc.json", "r01_hbm_traffic_pmc.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    traffic = {k: v["hbm_total"] for k, v in json.load(f)["per_clip_bytes"].items()}
                traffic_file = "profiles/" + name
                break
            except (OSError, KeyError, ValueError):
                continue
        default_cfg = (a.backbone, T, H, Wd) == ("video-swin-t", 8, 360, 640)
        split_on = {"win_attn3d": hot_ops.k1_split_enabled(), "linear_split": hot_ops.split_enabled(),
                    "ws_linear": hot_ops.k13_split_enabled(), "mlp_split": True, "xs_linear": True}
        desc = {"win_attn3d": "soc_win_attn3d_f32 (K1, 3-D shifted-window attention)",
                "linear_split": "soc_linear_split_f32 (K20, tiled linear layers)",
                "ws_linear": "soc_ws_linear_f32 (K13 / K13b, weight-stationary linear layers)",
                "xs_linear": "soc_xs_linear_f32 (K24, x-stationary linear layers: rows split once, weights streamed)",
                "mlp_split": "soc_mlp_split_f32 (K23: LayerNorm + linear + activation + linear + residual (+ LayerNorm) in one "
                             "launch, hidden layer in registers -- the Video-Swin MLPs of stages 0-2 and the encoder's feed-forward "
                             "blocks)"}
        stats_rows = {"win_attn3d": "win_attn3d_stream_kernel<false|true>", "linear_split": "linear_split_kernel<...>",
                      "ws_linear": "ws_linear_split_kernel<...> (+ ws_linear_kernel<...>)", "xs_linear": "xs_linear_kernel<...>",
                      "mlp_split": "mlp_split_kernel<...> (+ mlp_reduce_kernel)"}

        def ceiling(fam, shapes):
            """Time the launches cannot beat: per launch max(FLOPs / matrix-core peak, algorithmic bytes / achievable HBM
            rate).  The matrix-core peak of a kernel that runs its f32 products as six bf16 MFMAs (exact three-way operand
            split) is the dense bf16 peak / 6 = 417 TFLOP/s in algorithmic f32 FLOPs; of the f32-input MFMA forms 157.3."""
            mf = (PEAK_BF16_MFMA_TFLOPS / 6.0 if split_on[fam] else PEAK_F32_MFMA_TFLOPS) * 1e12
            t_m = sum(sh["launches"] * sh["flop"] / mf for sh in shapes)
            t_h = sum(sh["launches"] * sh["bytes"] / (HBM_ACHIEVABLE_GBS * 1e9) for sh in shapes)
            t_c = sum(sh["launches"] * max(sh["flop"] / mf, sh["bytes"] / (HBM_ACHIEVABLE_GBS * 1e9)) for sh in shapes)
            return t_c, t_m, t_h, mf / 1e12

        def vector_share(fam):
            """VALU cycles per matrix-pipe cycle of the family's kernels, from the committed rocprofv3 counter passes
            (SQ_INSTS_VALU x 2 clk / (SQ_INSTS_MFMA x 16.7 clk); v_exp and packed forms counted as one instruction).  On a
            gfx950 SIMD the two do not overlap beside v_mfma_f32_16x16x32_bf16 (tools/microbench/mfma_valu_roles.hip), so
            1 / (1 + share) bounds such a kernel below the 417 TFLOP/s ceiling before any LDS / barrier stall; beside the
            32x32x16 form (K1 since round 6) about four vector instructions per MFMA do (tools/microbench/mfma32_valu.hip)."""
            src = {"win_attn3d": ("r06_k1_pmc.json", lambda d: d["counters"]["k1s0_stream_x10"]),
                   "mlp_split": ("r04_k23enc_counters.json", lambda d: d["k23enc"]),
                   "xs_linear": ("r04_k24_counters.json", lambda d: d["counters"]["k24qkv2"])}.get(fam)
            if src is None:
                return None
            try:
                with open(os.path.join(ROOT, "profiles", src[0])) as fh:
                    c = src[1](json.load(fh))
                mfma_clk = 33.4 if fam == "win_attn3d" else 16.7        # K1 (round 6) issues v_mfma_f32_32x32x16_bf16: twice the cycles each
                return {"valu_cycles_per_mfma_cycle": round(2.0 * c["SQ_INSTS_VALU"]["mean_per_launch"]
                                                            / (mfma_clk * c["SQ_INSTS_MFMA"]["mean_per_launch"]), 3),
                        "source": "profiles/" + src[0],
                        "note": "matrix-pipe and vector-ALU time add on a gfx950 SIMD (DESIGN.md section 3, "
                                "tools/microbench/mfma_valu_roles.hip): 1 / (1 + this) bounds frac_of_ceiling"}
            except (OSError, KeyError, ValueError):
                return None

        def kernel_label(fam, launches):
            name, tag = desc[fam].split(" (")[0], desc[fam].split(" (")[1].split(",")[0].split(":")[0].rstrip(")")
            return (f"{name} ({tag}): all {launches} launches of one replay"
                    + (f" = {per} clips per head launch" if per > 1 else " = one forward"))

        blocks = {}
        for fam, f in families.items():
            flop = sum(sh["launches"] * sh["flop"] for sh in f["shapes"])
            byts = sum(sh["launches"] * sh["bytes"] for sh in f["shapes"])
            t_c, t_m, t_h, mf = ceiling(fam, f["shapes"])
            sec = f["ms"] * 1e-3
            mfma_bound = t_m >= t_h
            ach = flop / sec / 1e12 if mfma_bound else byts / sec / 1e9
            for sh in f["shapes"]:
                sh["tflops"] = round(sh["flop"] / sh["us"] / 1e6, 1)
                sh["gbs"] = round(sh["bytes"] / sh["us"] / 1e3, 1)
                sh["frac_of_ceiling"] = round(max(sh["flop"] / (mf * 1e12), sh["bytes"] / (HBM_ACHIEVABLE_GBS * 1e9)) / (sh["us"] * 1e-6), 3)
            blocks[fam] = {
                "kernel": kernel_label(fam, f["launches"]),
                "kernel_long": desc[fam], "launches": f["launches"], "clips_per_replay": per,
                "bound": "mfma" if mfma_bound else "hbm",
                "achieved": ach, "peak": mf if mfma_bound else PEAK_HBM_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "peak_basis": ("HBM3E 8 TB/s" if not mfma_bound else
                               "2.5 PFLOP/s dense bf16 MFMA / 6 bf16 products per f32 product (exact 3-way operand split); the f32-input "
                               "MFMA peak is 157.3" if split_on[fam] else "157.3 TFLOP/s f32-input MFMA"),
                "frac": ach / (mf if mfma_bound else PEAK_HBM_GBS),
                "frac_of_ceiling": t_c / sec,
                "ceiling": "per launch max(algorithmic FLOPs / matrix-core peak, algorithmic bytes / 6.3 TB/s achievable HBM rate), "
                           "summed over the launches, divided by the measured time; the matrix-core peak is "
                           + ("the dense bf16 MFMA peak / 6 = 417 TFLOP/s of algorithmic f32 FLOPs: every f32 product runs as six "
                              "bf16 MFMA products (exact three-way operand split)" if split_on[fam] else
                              "the f32-input MFMA peak, 157.3 TFLOP/s"),
                "ceiling_ms_per_clip": 1e3 * t_c / per, "mfma_ms_at_peak": 1e3 * t_m / per, "hbm_ms_at_6_3_TBs": 1e3 * t_h / per,
                "vs_f32_mfma_peak": flop / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "vector_share": vector_share(fam) if split_on[fam] else None,
                "algorithmic_tflops": flop / sec / 1e12, "algorithmic_gbs": byts / sec / 1e9,
                "traffic": traffic.get(fam) if default_cfg else None,
                "traffic_unit": f"HBM bytes per clip, rocprofv3 PMC of the one-clip eager forward, {traffic_file}",
                "algorithmic_flop_per_clip": flop / per, "algorithmic_bytes_per_clip": byts / per,
                "avg_launch_us": 1e3 * f["ms"] / f["launches"], "ms_per_clip": f["ms"] / per, "per_shape": f["shapes"],
                "measured": f"HIP events on the launch stream around {K1_REPS} back-to-back replays of this family's launches of one "
                            "forward (the forward's own activations and weights), right after the timed region",
                "per_launch_event_pairs_ms_per_clip": prof[fam]["ms"] / prof_clips if fam in prof else None,
                "source": f"HIP events in bench.py; rocprofv3 --kernel-trace --stats of this command: profiles/{PROFILE_TAG}_bench_kernel_stats.csv "
                          f"rows {stats_rows[fam]}"}
        if blocks:      # the roofline object is the kernel family with the largest share of a clip; the others follow
            order = sorted(blocks, key=lambda n: -blocks[n]["ms_per_clip"])
            line["roofline"] = blocks[order[0]]
            for n in order[1:]:
                line["roofline_" + n] = blocks[n]
        other = {}
        for name in ("msda_fwd", "xattn", "dyn_mask", "add_layernorm", "groupnorm_tokens", "patch_merge_layernorm"):
            if name in prof:
                r = prof[name]
                gbs = r["work"] / (r["ms"] * 1e-3) / 1e9
                other[name] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": gbs / PEAK_HBM_GBS, "avg_launch_us": 1e3 * r["ms"] / r["launches"],
                               "ms_per_clip": r["ms"] / prof_clips,
                               "algorithmic_bytes_per_clip": r["work"] / prof_clips,
                               "traffic": traffic.get(name) if default_cfg else None}
        if "msda_fwd" in other:
            # K2's binding unit is not HBM: the CU's vector-memory (texture) path, 64 B per clock -- measured busy share from
            # the committed counter passes
            try:
                if H >= 720:
                    k2_file, k2_key = "r05_k2_counters.json", "k2_fused_720p"
                else:                       # round 6: counters at the frame count of a launch group (80 frames) and of one clip
                    k2_file, k2_key = "r06_k2_counters.json", "k2_fused_360p_x10" if per > 1 else "k2_fused_360p_x1"
                with open(os.path.join(ROOT, "profiles", k2_file)) as fh:
                    k2 = json.load(fh)[k2_key]["derived"]
                other["msda_fwd"]["binding_unit"] = ("vector-memory (texture) path of the CUs, 64 B/clk each: busy "
                                                     f"{k2['ta_busy_share']:.2f} of the launch (profiles/{k2_file})")
                other["msda_fwd"]["ta_busy_share"] = round(k2["ta_busy_share"], 3)
            except (OSError, KeyError, ValueError):
                pass
        line["roofline_other"] = other
        # every hand-written kernel of the forward (HIP-event time of the instrumented eager pass)
        from neurips2023_soc_amd.graph_runner import switches_set
        line["switches"] = switches_set()          # diagnostic environment switches in effect (empty in a default run)
        line["kernels_per_forward"] = {"hand_written_launches": round(sum(r["launches"] for r in prof.values()) / prof_clips, 1),
                                       "source": "instrumented eager pass (hot_ops.profile_*); library launches not counted"}
        line["kernel_ms_per_clip"] = {name: {"launches_per_clip": r["launches"] / prof_clips, "ms": r["ms"] / prof_clips}
                                      for name, r in sorted(prof.items())}

        # What the timed region itself produced, against the reference-generated golden of this configuration (committed
        # data, not /root/reference): record i of rank 0 is pool clip i = seed 1 + i; the golden names its seed.
        golden_name = golden_cfg(a.backbone, T, H, Wd)[0]
        golden_path = os.path.join(ROOT, "tests", "golden", golden_name) if golden_name else None
        g_rec = 0
        if golden_path and os.path.exists(golden_path):
            import numpy as np
            with np.load(golden_path) as z:
                g = {k: z[k] for k in ("selected_query", "selected_masks", "pred_cls", "cfg")}
            g_rec = int(g["cfg"][0]) - 1
            assert int(g["cfg"][0]) == golden_cfg(a.backbone, T, H, Wd)[1]
            assert tuple(int(v) for v in g["cfg"][1:]) == (T, H, Wd, L) and 0 <= g_rec < timed_records.shape[0], g["cfg"]
            q, cls, masks = CP.unpack_record(timed_records[g_rec], T, Q, hm, wm)
            want = torch.from_numpy(g["selected_masks"]).reshape(T, hm, wm)
            flip = (masks > 0) != (want > 0)
            line["parity"] = {
                "timed_path_checked": f"record {g_rec} of the timed region vs tests/golden/{golden_name} (reference output)",
                "timed_path_selected_query": q, "timed_path_selected_query_ref": int(g["selected_query"]),
                "timed_path_mask_logit_max_abs_diff": float((masks - want).abs().max()),
                "timed_path_pred_cls_max_abs_diff": float((cls - torch.from_numpy(g["pred_cls"]).reshape(T, Q)).abs().max()),
                "timed_path_thresholded_mask_flips": int(flip.sum()),
                "timed_path_max_abs_ref_logit_at_flips": float(want[flip].abs().max()) if bool(flip.any()) else 0.0,
                "timed_path_pixels": flip.numel(),
                "flip_window": FLIP_WINDOW}
            assert q == int(g["selected_query"]) and line["parity"]["timed_path_mask_logit_max_abs_diff"] < 1e-3, \
                line["parity"]
            assert line["parity"]["timed_path_max_abs_ref_logit_at_flips"] < FLIP_WINDOW, line["parity"]
            # the records of clips 1..3 have no reference golden; they must at least be finite and distinct
            assert bool(torch.isfinite(timed_records).all())

        if world == 1 and not a.no_cpu_baseline:
            from oracle import soc_oracle as O
            enc = O.build_text_encoder(sd)
            ones = torch.ones_like(ids_cpu)
            granted, quota, avail = granted_cpus()
            phys, model_name = _cpu_topology()
            proxy = W.synthetic_clip(7, 3, 250, 300)

            def timed_forwards(n_threads, clip, size, reps):
                torch.set_num_threads(n_threads)
                ts, ref = [], None
                for r in range(reps + 1):       # first one is the warm-up
                    t1 = time.perf_counter()
                    ref = O.soc_forward(sd, clip, ids_cpu, ones, size, backbone=a.backbone, text_encoder=enc)
                    ts.append(time.perf_counter() - t1)
                ts = sorted(ts[1:])
                return ts[len(ts) // 2], ref

            runs, ref, best = [], None, None
            proxy8 = None
            # Configurations other than the headline one (Swin-B: 11 s per clip at 360p, 70 s at 720p on these hosts) keep the
            # CPU leg bounded: one thread count (what the quota grants), one warm-up-free forward of the golden's clip.
            light = not default_cfg
            # n = 8 (comparable with SURVEY section 6, measured on the real reference with 8 cores) and n = the CPUs the
            # box really grants (cgroup quota; SURVEY 8d "all physical host cores" as far as the container has them)
            for n in ([granted] if light else sorted({min(8, granted), granted})):
                tp, _ = timed_forwards(n, proxy, (250, 300), 1)
                proxy8 = proxy8 or tp
                entry = {"threads": n, "proxy_T3_250x300_s": tp}
                # the pod may expose far more logical CPUs than its cgroup grants (256 threads: 373 s per clip in
                # round 1): a thread count whose small proxy is already >2x slower than 8 threads is not run at size
                if light:
                    torch.set_num_threads(n)
                    t1 = time.perf_counter()
                    ref = O.soc_forward(sd, clips_cpu[g_rec], ids_cpu, ones, (H, Wd), backbone=a.backbone, text_encoder=enc)
                    med = time.perf_counter() - t1
                    entry.update(seconds_per_clip_single_cold_forward=med, clips_per_s=1.0 / med)
                    best = (n, med)
                elif tp <= 2.0 * proxy8:
                    med, ref = timed_forwards(n, clips_cpu[0], (H, Wd), 3)
                    entry.update(seconds_per_clip_median_of_3=med, clips_per_s=1.0 / med)
                    if best is None or med < best[1]:
                        best = (n, med)
                else:
                    entry["skipped"] = "oversubscribed: proxy > 2x the 8-thread proxy"
                runs.append(entry)
            line["cpu_baseline"] = {"value": 1.0 / best[1], "unit": "clips/s", "cores": best[0], "kind": "port",
                                    "cpu_model": model_name, "physical_cores_visible": phys, "logical_cpus": avail,
                                    "cgroup_cpu_quota": quota, "granted_cpus": granted, "runs": runs,
                                    "sample": ("same workload (oracle/soc_oracle.py, torch-CPU fp32): ONE forward, no warm-up, at the "
                                               "CPU count the cgroup quota grants (bounded: this is not the headline configuration)")
                                              if light else
                                              "same workload (oracle/soc_oracle.py, torch-CPU fp32): 1 warm-up + 3 timed "
                                              "forwards, median, at 8 threads and at the CPU count the cgroup quota grants"}
            d = (timed_records[g_rec][1 + T * Q:].view(T, hm, wm) - P.select_trajectory(ref)[1]).abs().max().item()
            line.setdefault("parity", {})["timed_path_mask_logit_max_abs_diff_vs_cpu_oracle"] = d
            # the other clips of the pool have no reference golden: one oracle forward each checks their timed records too
            # (selected query, its mask logits within the north_star tolerance, flips only inside fp32 noise of zero)
            others = []
            torch.set_num_threads(best[0])
            for i in ([] if light else range(1, min(n_pool, a.steps, 4))):
                ref_i = O.soc_forward(sd, clips_cpu[i], ids_cpu, ones, (H, Wd), backbone=a.backbone, text_encoder=enc)
                q_ref, m_ref = P.select_trajectory(ref_i)[:2]
                q_i, _, m_i = CP.unpack_record(timed_records[i], T, Q, hm, wm)
                flip_i = (m_i > 0) != (m_ref > 0)
                others.append({"record": i, "selected_query": q_i, "selected_query_oracle": int(q_ref),
                               "mask_logit_max_abs_diff": float((m_i - m_ref).abs().max()),
                               "thresholded_mask_flips": int(flip_i.sum()),
                               "max_abs_oracle_logit_at_flips": float(m_ref[flip_i].abs().max()) if bool(flip_i.any()) else 0.0})
                assert q_i == int(q_ref) and others[-1]["mask_logit_max_abs_diff"] < 1e-3 \
                    and others[-1]["max_abs_oracle_logit_at_flips"] < FLIP_WINDOW, others[-1]
            line["parity"]["timed_path_other_records_vs_cpu_oracle"] = others
            got = step(g_rec)
            torch.cuda.synchronize()
            d = (got["pred_masks"].cpu() - ref["pred_masks"]).abs().max().item()
            flip = (got["pred_masks"].cpu() > 0) != (ref["pred_masks"] > 0)
            line["parity"].update({"eager_all_queries_mask_logit_max_abs_diff_vs_cpu_oracle": d,
                                   "max_abs_logit": ref["pred_masks"].abs().max().item(),
                                   "eager_all_queries_thresholded_mask_flips": int(flip.sum()),
                                   "eager_all_queries_max_abs_ref_logit_at_flips":
                                       float(ref["pred_masks"][flip].abs().max()) if bool(flip.any()) else 0.0,
                                   "eager_all_queries_pixels": flip.numel()})
        emit(compact_line(line), line, a.detail)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
