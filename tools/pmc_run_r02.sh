#!/bin/bash
# Round-2 evidence run on the MI355X box (gpurun): everything lands under gpurun_out/r02/, the summaries are then
# copied into profiles/.  Counters are collected in their own passes (--pmc with --kernel-trace only), the program
# directly after `--`.
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
P=gpurun_out/r02
mkdir -p $P
# ---- (a) the headline command under the kernel tracer
python3 bench.py > $P/bench_r02_n1.json 2> $P/bench_r02_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $P/trace_bench.json 2> $P/trace_bench.err
T=$(ls $P/trace/*/*kernel_trace.csv | head -1)
cp $(ls $P/trace/*/*kernel_stats.csv | head -1) $P/r02_bench_kernel_stats.csv
python3 tools/analyze_trace.py $T --top 30 > $P/r02_forward_breakdown.txt
python3 tools/timeline.py $T > $P/r02_timeline.txt
python3 tools/launch_sequence.py $T > $P/r02_launch_sequence.txt
# ---- (b) HBM traffic per kernel and clip: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline > $P/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline > $P/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $(ls $P/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $P/pmc_write/*/*counter_collection.csv | head -1) > $P/r02_hbm_traffic_pmc.json
# ---- (c) K1 per stage: matrix-pipe / VALU / LDS counters
for st in 0 1 2 3; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/k1s${st}_sq -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/k1s${st}_lds -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_lds.log 2>&1
  python3 tools/pmc_agg.py --kernels "k1_stage$st=win_attn3d_full_kernel" -- $P/k1s${st}_sq $P/k1s${st}_lds > $P/k1s${st}_counters.json
done
# ---- (d) K2 (plain fused launch): where the gather is served from
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum -d $P/k2_tcp -- python3 tools/k2_probe.py 12 plain > $P/k2_tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $P/k2_tcc -- python3 tools/k2_probe.py 12 plain > $P/k2_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $P/k2_sq -- python3 tools/k2_probe.py 12 plain > $P/k2_sq.log 2>&1
python3 tools/pmc_agg.py --kernels "k2_fused=msda_fused_tiles_kernel" -- $P/k2_tcp $P/k2_tcc $P/k2_sq > $P/k2_counters.json
# ---- (e) K4: vector issue vs wave residency
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $P/k4_sq -- python3 tools/run_kernel.py dyn 12 > $P/k4_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_WR -d $P/k4_sq2 -- python3 tools/run_kernel.py dyn 12 > $P/k4_sq2.log 2>&1
python3 tools/pmc_agg.py --kernels "k4=dyn_mask_kernel" -- $P/k4_sq $P/k4_sq2 > $P/k4_counters.json
# ---- (f) stage times of the replay and the other named configs
python3 tools/head_probe.py > $P/head_probe.txt 2>&1
python3 bench.py --no-cpu-baseline --no-pipeline > $P/bench_r02_n1_one_clip_per_replay.json 2> /dev/null
python3 bench.py --no-cpu-baseline --backbone video-swin-b > $P/bench_r02_swinb_360p.json 2> /dev/null
python3 bench.py --no-cpu-baseline --backbone video-swin-b --height 720 --width 1280 --steps 10 > $P/bench_r02_swinb_720p.json 2> /dev/null
python3 tools/k1_probe.py > $P/k1_probe_time.txt 2>&1
python3 tools/k1_probe.py --stamps 0 2 > $P/k1_probe_stamps.txt 2>&1
python3 tools/k2_probe.py 50 plain > $P/k2_probe_time.txt 2>&1
# keep the merge small
find $P -name "*kernel_trace.csv" -size +8M -delete
find $P -name "*.db" -delete
du -sh $P
