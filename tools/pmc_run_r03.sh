#!/bin/bash
# Round-3 evidence run on the MI355X box (gpurun): everything lands under gpurun_out/r03/, the summaries are then
# copied into profiles/.  Counters are collected in their own passes (--pmc with --kernel-trace only), the program
# directly after `--`.
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
P=gpurun_out/r03
mkdir -p $P
# ---- (a) the headline command, plain and under the kernel tracer
python3 bench.py > $P/bench_r03_n1.json 2> $P/bench_r03_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $P/trace_bench.json 2> $P/trace_bench.err
T=$(ls $P/trace/*/*kernel_trace.csv | head -1)
cp $(ls $P/trace/*/*kernel_stats.csv | head -1) $P/r03_bench_kernel_stats.csv
python3 tools/analyze_trace.py $T --top 30 > $P/r03_forward_breakdown.txt
python3 tools/timeline.py $T > $P/r03_timeline.txt
python3 tools/launch_sequence.py $T > $P/r03_launch_sequence.txt
# ---- (b) HBM traffic per kernel and clip: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-stream --no-f32-pass > $P/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-stream --no-f32-pass > $P/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $(ls $P/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $P/pmc_write/*/*counter_collection.csv | head -1) > $P/r03_hbm_traffic_pmc.json
# ---- (c) K1 per stage, split (bf16 matrix cores) and f32 form: matrix-pipe / VALU / LDS counters
for form in split f32; do
  if [ $form = f32 ]; then export SOC_MATMUL=f32; else unset SOC_MATMUL; fi
  for st in 0 1 2 3; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/k1s${st}_${form}_sq -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_${form}_sq.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/k1s${st}_${form}_lds -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_${form}_lds.log 2>&1
    python3 tools/pmc_agg.py --kernels "k1_stage${st}_${form}=win_attn3d_" -- $P/k1s${st}_${form}_sq $P/k1s${st}_${form}_lds > $P/k1s${st}_${form}_counters.json
  done
done
unset SOC_MATMUL
# ---- (d) K20 at three call sites
for site in k20ffn k20qkv1 k20fc1s2; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/${site}_sq -- python3 tools/run_kernel.py $site 12 > $P/${site}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/${site}_lds -- python3 tools/run_kernel.py $site 12 > $P/${site}_lds.log 2>&1
  python3 tools/pmc_agg.py --kernels "${site}=linear_split_kernel" -- $P/${site}_sq $P/${site}_lds > $P/${site}_counters.json
done
# ---- (d2) K13b at three call sites and K22 (one round of 32 768 rows)
for site in k13qkv0 k13fc1s0 k13qkv2 k22; do
  pat=ws_linear_split_kernel; [ $site = k22 ] && pat=ffn_split_kernel
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/${site}_sq -- python3 tools/run_kernel.py $site 12 > $P/${site}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/${site}_lds -- python3 tools/run_kernel.py $site 12 > $P/${site}_lds.log 2>&1
  python3 tools/pmc_agg.py --kernels "${site}=${pat}" -- $P/${site}_sq $P/${site}_lds > $P/${site}_counters.json
done
# ---- (e) K2 (plain fused launch): where the gather is served from
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum -d $P/k2_tcp -- python3 tools/k2_probe.py 12 plain > $P/k2_tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $P/k2_tcc -- python3 tools/k2_probe.py 12 plain > $P/k2_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $P/k2_sq -- python3 tools/k2_probe.py 12 plain > $P/k2_sq.log 2>&1
python3 tools/pmc_agg.py --kernels "k2_fused=msda_fused_tiles_kernel" -- $P/k2_tcp $P/k2_tcc $P/k2_sq > $P/k2_counters.json
# ---- (f) stage times of the replay, the other named configs, probes
python3 tools/head_probe.py > $P/head_probe.txt 2>&1
python3 tools/gemm_sites.py 5 > $P/gemm_sites.txt 2>&1
python3 bench.py --no-cpu-baseline --no-pipeline --no-stream --no-f32-pass > $P/bench_r03_n1_one_clip_per_replay.json 2> /dev/null
python3 bench.py --no-cpu-baseline --backbone video-swin-b --no-stream > $P/bench_r03_swinb_360p.json 2> /dev/null
python3 bench.py --no-cpu-baseline --backbone video-swin-b --height 720 --width 1280 --steps 10 --no-stream > $P/bench_r03_swinb_720p.json 2> /dev/null
python3 tools/k1_probe.py > $P/k1_probe_time.txt 2>&1
python3 tools/k2_probe.py 50 plain > $P/k2_probe_time.txt 2>&1
./tools/experiments/_build/pk_mfma_probe 40 > $P/pk_mfma_probe.txt 2>&1
python3 tools/experiments/k20_vs_dynmask.py 300 none k20 k20_s0 k1 k13 > $P/k4_beside_kernels.txt 2>&1
python3 tools/experiments/k22_time.py > $P/k22_time.txt 2>&1
python3 tools/experiments/k13b_time.py > $P/k13b_time.txt 2>&1
SOAK_TRACE=1 python3 tools/experiments/soak_sites.py 6000 2> /dev/null | tail -1 > $P/soak_trace.json
# keep the merge small
find $P -name "*kernel_trace.csv" -size +8M -delete
find $P -name "*.db" -delete
du -sh $P
