#!/bin/bash
# Round-5 evidence run on the MI355X box (gpurun): everything lands under gpurun_out/r05/, the summaries are then copied
# into profiles/.  Counters are collected in their own passes (--pmc with --kernel-trace only), the program directly after
# `--`.  Parts: trace | traffic | k2 | swinb | rest (default: all).
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
GROUP=10                                # bench.py's default at the driver's 20 clips: two launch groups of ten (default_pipeline()); the trace / PMC tools count per group
export SOC_TRACE_CLIPS_PER_GROUP=$GROUP
P=gpurun_out/r05
mkdir -p $P
PARTS=${1:-trace traffic k2 swinb rest}
for part in $PARTS; do
case $part in
trace)
  # ---- (a) the headline command, plain and under the kernel tracer
  python3 bench.py --steps 20 --warmup 5 --detail $P/bench_r05_n1_detail.json > $P/bench_r05_n1.json 2> $P/bench_r05_n1.err
  python3 bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-stream --no-f32-pass --detail $P/bench_r05_n1_200steps_detail.json > $P/bench_r05_n1_200steps.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 60 --warmup 10 --pipeline group$GROUP --no-cpu-baseline --detail $P/trace_bench_detail.json > $P/trace_bench.json 2> $P/trace_bench.err
  T=$(ls $P/trace/*/*kernel_trace.csv | head -1)
  cp $(ls $P/trace/*/*kernel_stats.csv | head -1) $P/r05_bench_kernel_stats.csv
  python3 tools/analyze_trace.py $T --top 30 > $P/r05_forward_breakdown.txt
  python3 tools/analyze_trace.py $T --segments > $P/r05_trace_segments.txt
  python3 tools/timeline.py $T > $P/r05_timeline.txt
  python3 tools/launch_sequence.py $T > $P/r05_launch_sequence.txt
  # the one-clip-per-launch pipeline of rounds 1-4 under the same tracer (same box)
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_onegraph -- python3 bench.py --pipeline one-graph --steps 20 --warmup 3 --no-cpu-baseline --no-stream --no-f32-pass --detail $P/trace_onegraph_detail.json > $P/trace_onegraph.json 2> $P/trace_onegraph.err
  cp $(ls $P/trace_onegraph/*/*kernel_stats.csv | head -1) $P/r05_bench_kernel_stats_single_clip.csv
  SOC_TRACE_CLIPS_PER_GROUP=1 python3 tools/timeline.py $(ls $P/trace_onegraph/*/*kernel_trace.csv | head -1) > $P/r05_timeline_single_clip.txt
  SOC_TRACE_CLIPS_PER_GROUP=1 python3 tools/analyze_trace.py $(ls $P/trace_onegraph/*/*kernel_trace.csv | head -1) --top 30 > $P/r05_forward_breakdown_single_clip.txt
  ;;
traffic)
  # ---- (b) HBM traffic per kernel and clip: FETCH_SIZE and WRITE_SIZE in separate passes
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --eager --steps $GROUP --warmup $GROUP --no-cpu-baseline --no-stream --no-f32-pass --detail $P/pmc_fetch_detail.json > $P/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --eager --steps $GROUP --warmup $GROUP --no-cpu-baseline --no-stream --no-f32-pass --detail $P/pmc_write_detail.json > $P/pmc_write.log 2>&1
  python3 tools/pmc_traffic.py $(ls $P/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $P/pmc_write/*/*counter_collection.csv | head -1) $GROUP > $P/r05_hbm_traffic_pmc.json
  ;;
k2)
  # ---- (c) K2 (fused multi-scale deformable attention): where the gather is served from, at 360p and at 720p (config 4)
  for geo in 360p 720p; do
    python3 tools/k2_probe.py 50 $geo > $P/k2_${geo}_time.txt 2>&1
    rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum -d $P/k2_${geo}_tcp -- python3 tools/k2_probe.py 12 $geo > $P/k2_${geo}_tcp.log 2>&1
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum -d $P/k2_${geo}_tcc -- python3 tools/k2_probe.py 12 $geo > $P/k2_${geo}_tcc.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD -d $P/k2_${geo}_sq -- python3 tools/k2_probe.py 12 $geo > $P/k2_${geo}_sq.log 2>&1
    python3 tools/pmc_agg.py --kernels "k2_fused_${geo}=msda_fused_tiles_kernel" -- $P/k2_${geo}_tcp $P/k2_${geo}_tcc $P/k2_${geo}_sq > $P/k2_${geo}_counters.json
  done
  ;;
swinb)
  # ---- (d) BASELINE configs 4 / 5: Video-Swin-B at 720p and at 360p -- bench line with cpu_baseline + golden parity, kernel stats, HBM traffic
  python3 bench.py --backbone video-swin-b --steps 20 --warmup 5 --no-stream --detail $P/bench_r05_swinb_360p_detail.json > $P/bench_r05_swinb_360p.json 2> $P/bench_r05_swinb_360p.err
  python3 bench.py --backbone video-swin-b --height 720 --width 1280 --steps 10 --no-stream --detail $P/bench_r05_swinb_720p_detail.json > $P/bench_r05_swinb_720p.json 2> $P/bench_r05_swinb_720p.err
  for geo in 360p 720p; do
    # 720p runs pairs (bench.py's default above 360x640), 360p groups of $GROUP
    if [ $geo = 720p ]; then G="--height 720 --width 1280 --steps 6"; PERG=2; E="--steps 8 --warmup 4"; else G="--steps 40 --pipeline group$GROUP"; PERG=$GROUP; E="--steps $GROUP --warmup $GROUP"; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_swinb_$geo -- python3 bench.py --backbone video-swin-b $G --warmup 2 --no-cpu-baseline --no-stream --no-f32-pass --detail $P/trace_swinb_${geo}_detail.json > $P/trace_swinb_$geo.json 2> $P/trace_swinb_$geo.err
    cp $(ls $P/trace_swinb_$geo/*/*kernel_stats.csv | head -1) $P/r05_swinb_${geo}_kernel_stats.csv
    SOC_TRACE_CLIPS_PER_GROUP=$PERG python3 tools/analyze_trace.py $(ls $P/trace_swinb_$geo/*/*kernel_trace.csv | head -1) --top 25 > $P/r05_swinb_${geo}_forward_breakdown.txt
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch_swinb_$geo -- python3 bench.py --backbone video-swin-b $G --eager $E --no-cpu-baseline --no-stream --no-f32-pass --detail $P/x.json > $P/pmc_fetch_swinb_$geo.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write_swinb_$geo -- python3 bench.py --backbone video-swin-b $G --eager $E --no-cpu-baseline --no-stream --no-f32-pass --detail $P/x.json > $P/pmc_write_swinb_$geo.log 2>&1
    python3 tools/pmc_traffic.py $(ls $P/pmc_fetch_swinb_$geo/*/*counter_collection.csv | head -1) $(ls $P/pmc_write_swinb_$geo/*/*counter_collection.csv | head -1) $PERG > $P/r05_swinb_${geo}_hbm_traffic_pmc.json
  done
  ;;
counters)
  # ---- (c2) K23 / K1 / K24 at the shapes of a launch group ($GROUP clips) and of one clip: matrix-pipe / VALU / LDS / wait counters
  for clips in $GROUP 1; do
    for site in k23enc k23s2 k1s0 k1s2 k24qkv2 k24qkv3; do
      case $site in k23*) pat=mlp_split_kernel;; k1*) pat=win_attn3d_;; *) pat=xs_linear_kernel;; esac
      tag=${site}_x${clips}
      rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/${tag}_sq -- python3 tools/run_kernel.py $site 12 $clips > $P/${tag}_sq.log 2>&1
      rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/${tag}_lds -- python3 tools/run_kernel.py $site 12 $clips > $P/${tag}_lds.log 2>&1
      python3 tools/pmc_agg.py --kernels "${tag}=${pat}" -- $P/${tag}_sq $P/${tag}_lds > $P/${tag}_counters.json
    done
  done
  python3 - <<'PY' > gpurun_out/r05/r05_group_counters.json
import glob, json, os
out = {"command": "tools/pmc_run_r05.sh counters: rocprofv3 --kernel-trace --pmc <8 SQ counters> | <LDS / co-execution counters, GRBM_GUI_ACTIVE> "
                  "(two separate passes) -- python3 tools/run_kernel.py <site> 12 <clips per launch group>; tools/pmc_agg.py: mean per launch.  "
                  "matrix_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128); wait_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"}
for f in sorted(glob.glob("gpurun_out/r05/*_x*_counters.json")):
    d = json.load(open(f))
    for k, c in d.items():
        if not isinstance(c, dict) or "SQ_INSTS_MFMA" not in c:
            continue
        g = c["GRBM_GUI_ACTIVE"]["mean_per_launch"]
        c["derived"] = {"matrix_pipe_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (g * 128),
                        "wait_share": c["SQ_WAIT_INST_ANY"]["mean_per_launch"] / c["SQ_WAVE_CYCLES"]["mean_per_launch"],
                        "valu_per_mfma": c["SQ_INSTS_VALU"]["mean_per_launch"] / c["SQ_INSTS_MFMA"]["mean_per_launch"]}
        out[k] = c
print(json.dumps(out, indent=1))
PY
  ;;
rest)
  # ---- (e) stage times of the replay, pipelines side by side on one box, probes
  python3 tools/head_probe.py > $P/head_probe.txt 2>&1
  python3 tools/experiments/pipeline_ab.py 40 3 > $P/pipeline_ab.txt 2>&1
  python3 tools/gemm_sites.py 5 > $P/gemm_sites.txt 2>&1
  python3 bench.py --no-cpu-baseline --no-pipeline --no-stream --no-f32-pass --detail $P/x.json > $P/bench_r05_n1_one_clip_per_replay.json 2> /dev/null
  python3 tools/experiments/batch2_probe.py > $P/batch2_probe.txt 2>&1
  ;;
esac
done
# keep the merge small
find $P -name "*kernel_trace.csv" -size +8M -delete
find $P -name "*counter_collection.csv" -size +8M -delete
find $P -name "*.db" -delete
rm -f $P/x.json
du -sh $P
