#!/bin/bash
# Round-6 evidence run on the MI355X box (gpurun): everything lands under gpurun_out/r06/, the summaries are then copied
# into profiles/.  Counters are collected in their own passes (--pmc with --kernel-trace only), the program directly after
# `--`.  Parts: trace | traffic | counters | k2 | png | swinb (default: all but swinb).
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
GROUP=10                                # bench.py's fixed launch group (DEFAULT_GROUP); the trace / PMC tools count per group
export SOC_TRACE_CLIPS_PER_GROUP=$GROUP
P=gpurun_out/r06
mkdir -p $P
PARTS=${1:-trace traffic counters k2 png}
for part in $PARTS; do
case $part in
trace)
  # ---- (a) the headline command, plain and under the kernel tracer
  python3 bench.py --steps 20 --warmup 5 --detail $P/bench_r06_n1_detail.json > $P/bench_r06_n1.json 2> $P/bench_r06_n1.err
  python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-stream --no-f32-pass --detail $P/bench_r06_n1_200steps_detail.json > $P/bench_r06_n1_200steps.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-single-pass --detail $P/trace_bench_detail.json > $P/trace_bench.json 2> $P/trace_bench.err
  T=$(ls $P/trace/*/*kernel_trace.csv | head -1)
  cp $(ls $P/trace/*/*kernel_stats.csv | head -1) $P/r06_bench_kernel_stats.csv
  python3 tools/analyze_trace.py $T --top 30 > $P/r06_forward_breakdown.txt
  python3 tools/timeline.py $T > $P/r06_timeline.txt
  python3 tools/launch_sequence.py $T > $P/r06_launch_sequence.txt
  ;;
traffic)
  # ---- (b) HBM traffic per kernel and clip at the launch-group shapes: FETCH_SIZE and WRITE_SIZE in separate passes
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --eager --steps $GROUP --warmup $GROUP --no-cpu-baseline --no-stream --no-f32-pass --detail $P/pmc_fetch_detail.json > $P/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --eager --steps $GROUP --warmup $GROUP --no-cpu-baseline --no-stream --no-f32-pass --detail $P/pmc_write_detail.json > $P/pmc_write.log 2>&1
  python3 tools/pmc_traffic.py $(ls $P/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $P/pmc_write/*/*counter_collection.csv | head -1) $GROUP > $P/r06_hbm_traffic_pmc.json
  ;;
counters)
  # ---- (c) K1 (streaming form, and the round-3 split form beside it) at the shapes of one clip and of a launch group
  for form in stream r3; do
    if [ $form = r3 ]; then export SOC_K1_FORM=r3; else unset SOC_K1_FORM; fi
    for clips in 1 $GROUP; do
      for site in k1s0 k1s2; do
        tag=${site}_${form}_x${clips}
        rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/${tag}_sq -- python3 tools/run_kernel.py $site 12 $clips > $P/${tag}_sq.log 2>&1
        rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/${tag}_lds -- python3 tools/run_kernel.py $site 12 $clips > $P/${tag}_lds.log 2>&1
        python3 tools/pmc_agg.py --kernels "${tag}=win_attn3d_" -- $P/${tag}_sq $P/${tag}_lds > $P/${tag}_counters.json
      done
    done
  done
  unset SOC_K1_FORM
  python3 - <<'PY' > gpurun_out/r06/r06_k1_pmc.json
import glob, json
out = {"command": "tools/pmc_run_r06.sh counters: rocprofv3 --kernel-trace --pmc <8 SQ counters> | <LDS / co-execution counters, GRBM_GUI_ACTIVE> "
                  "(two separate passes) -- python3 tools/run_kernel.py k1s<stage> 12 <clips per launch>; tools/pmc_agg.py: mean per launch.  "
                  "stream = win_attn3d_stream_kernel (round 6, split_arith 1), r3 = win_attn3d_split_kernel (round 3, SOC_K1_FORM=r3).  "
                  "matrix_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)", "counters": {}}
for f in sorted(glob.glob("gpurun_out/r06/k1s*_counters.json")):
    d = json.load(open(f))
    out["counters"].update(d.get("counters", d))
print(json.dumps(out, indent=1))
PY
  ;;
k2)
  # ---- (d) K2 at a launch group's 80 frames: HBM traffic back at the algorithmic bytes?  (round 5: 711 MB per clip vs 415)
  for clips in 1 $GROUP; do
    python3 tools/k2_probe.py 30 360p 1.0 $clips > $P/k2_x${clips}_time.txt 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/k2_x${clips}_fetch -- python3 tools/k2_probe.py 12 360p 1.0 $clips > $P/k2_x${clips}_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/k2_x${clips}_write -- python3 tools/k2_probe.py 12 360p 1.0 $clips > $P/k2_x${clips}_write.log 2>&1
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d $P/k2_x${clips}_tcc -- python3 tools/k2_probe.py 12 360p 1.0 $clips > $P/k2_x${clips}_tcc.log 2>&1
    python3 tools/pmc_agg.py --kernels "k2_fused_x${clips}=msda_fused_tiles_kernel" -- $P/k2_x${clips}_fetch $P/k2_x${clips}_write $P/k2_x${clips}_tcc > $P/k2_x${clips}_counters.json
  done
  ;;
swinb)
  # ---- (f) BASELINE configs 4 / 5: Video-Swin-B at 360p (groups of ten) and 720p (pairs) -- kernel stats and forward breakdown
  #      (the bench lines with cpu_baseline + golden parity: profiles/bench_r06_swinb_*.json)
  for geo in 360p 720p; do
    if [ $geo = 720p ]; then G="--height 720 --width 1280 --steps 6"; PERG=2; else G="--steps 40"; PERG=$GROUP; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_swinb_$geo -- python3 bench.py --backbone video-swin-b $G --warmup 2 --no-cpu-baseline --no-stream --no-f32-pass --no-single-pass --detail $P/trace_swinb_${geo}_detail.json > $P/trace_swinb_$geo.json 2> $P/trace_swinb_$geo.err
    cp $(ls $P/trace_swinb_$geo/*/*kernel_stats.csv | head -1) $P/r06_swinb_${geo}_kernel_stats.csv
    SOC_TRACE_CLIPS_PER_GROUP=$PERG python3 tools/analyze_trace.py $(ls $P/trace_swinb_$geo/*/*kernel_trace.csv | head -1) --top 25 > $P/r06_swinb_${geo}_forward_breakdown.txt
  done
  ;;
png)
  # ---- (e) files -> PNG through the dataset driver: launch groups of eight, ragged expression counts per video
  python3 tools/files_to_png.py --videos 64 --group 8 --ragged --out $P/r06_files_to_png.json > $P/files_to_png.log 2>&1
  python3 tools/files_to_png.py --videos 64 --group 8 --out $P/r06_files_to_png_3_per_video.json > $P/files_to_png3.log 2>&1
  ;;
esac
done
# only summaries travel back (gpurun merges at most 64 MiB): the raw traces / counter databases stay on the box
find $P -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
du -sh $P
ls -la $P | head -80
