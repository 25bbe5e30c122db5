#!/bin/bash
# Round-4 evidence run on the MI355X box (gpurun): everything lands under gpurun_out/r04/, the summaries are then copied
# into profiles/.  Counters are collected in their own passes (--pmc with --kernel-trace only), the program directly after
# `--`.  Parts: trace | traffic | k23 | k1k24 | rest (default: all).
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
P=gpurun_out/r04
mkdir -p $P
PARTS=${1:-trace traffic k23 k1k24 rest}
for part in $PARTS; do
case $part in
trace)
  # ---- (a) the headline command, plain and under the kernel tracer
  python3 bench.py > $P/bench_r04_n1.json 2> $P/bench_r04_n1.err
  python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-stream --no-f32-pass > $P/bench_r04_n1_200steps.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $P/trace_bench.json 2> $P/trace_bench.err
  T=$(ls $P/trace/*/*kernel_trace.csv | head -1)
  cp $(ls $P/trace/*/*kernel_stats.csv | head -1) $P/r04_bench_kernel_stats.csv
  python3 tools/analyze_trace.py $T --top 30 > $P/r04_forward_breakdown.txt
  python3 tools/timeline.py $T > $P/r04_timeline.txt
  python3 tools/launch_sequence.py $T > $P/r04_launch_sequence.txt
  ;;
traffic)
  # ---- (b) HBM traffic per kernel and clip: FETCH_SIZE and WRITE_SIZE in separate passes
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-stream --no-f32-pass > $P/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-stream --no-f32-pass > $P/pmc_write.log 2>&1
  python3 tools/pmc_traffic.py $(ls $P/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $P/pmc_write/*/*counter_collection.csv | head -1) > $P/r04_hbm_traffic_pmc.json
  ;;
k23)
  # ---- (c) K23 at its four call sites: matrix-pipe / VALU / LDS / wait counters
  for site in k23s0 k23s1 k23s2 k23enc; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/${site}_sq -- python3 tools/run_kernel.py $site 12 > $P/${site}_sq.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/${site}_lds -- python3 tools/run_kernel.py $site 12 > $P/${site}_lds.log 2>&1
    python3 tools/pmc_agg.py --kernels "${site}=mlp_split_kernel" -- $P/${site}_sq $P/${site}_lds > $P/${site}_counters.json
  done
  ;;
k1k24)
  # ---- (c2) K1 per stage (split form) and K24 at four call sites: the same two counter passes
  for st in 0 1 2 3; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/k1s${st}_sq -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_sq.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/k1s${st}_lds -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_lds.log 2>&1
    python3 tools/pmc_agg.py --kernels "k1_stage${st}_split=win_attn3d_" -- $P/k1s${st}_sq $P/k1s${st}_lds > $P/k1s${st}_counters.json
  done
  for site in k24qkv2 k24enc k24qkv3 k24vlf; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/${site}_sq -- python3 tools/run_kernel.py $site 12 > $P/${site}_sq.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/${site}_lds -- python3 tools/run_kernel.py $site 12 > $P/${site}_lds.log 2>&1
    python3 tools/pmc_agg.py --kernels "${site}=xs_linear_kernel" -- $P/${site}_sq $P/${site}_lds > $P/${site}_counters.json
  done
  ;;
rest)
  # ---- (d) stage times of the replay, the other named configs, probes
  python3 tools/head_probe.py > $P/head_probe.txt 2>&1
  python3 tools/gemm_sites.py 5 > $P/gemm_sites.txt 2>&1
  python3 bench.py --no-cpu-baseline --no-pipeline --no-stream --no-f32-pass > $P/bench_r04_n1_one_clip_per_replay.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --backbone video-swin-b --no-stream > $P/bench_r04_swinb_360p.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --backbone video-swin-b --height 720 --width 1280 --steps 10 --no-stream > $P/bench_r04_swinb_720p.json 2> /dev/null
  ;;
esac
done
# keep the merge small
find $P -name "*kernel_trace.csv" -size +8M -delete
find $P -name "*.db" -delete
du -sh $P
