set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
P=gpurun_out/pmc2
# ---- K2: plain vs encoder-form (one process runs both kernels)
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum -d $P/k2_tcp -- python3 tools/k2_probe.py 12 > $P/k2_tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $P/k2_tcc -- python3 tools/k2_probe.py 12 > $P/k2_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $P/k2_sq -- python3 tools/k2_probe.py 12 > $P/k2_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum -d $P/k2_lat -- python3 tools/k2_probe.py 12 > $P/k2_lat.log 2>&1
# ---- K1 per stage
for st in 0 1 2 3; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/k1s${st}_sq -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/k1s${st}_lds -- python3 tools/run_kernel.py k1s$st 12 > $P/k1s${st}_lds.log 2>&1
done
python3 tools/pmc_agg.py --kernels "k2_plain=msda_fwd_d32p4_kernel,true, false" "k2_encoder_form=msda_fwd_d32p4_kernel,true, true" -- $P/k2_tcp $P/k2_tcc $P/k2_sq $P/k2_lat > $P/k2_counters.json
for st in 0 1 2 3; do python3 tools/pmc_agg.py --kernels "k1_stage$st=win_attn3d_full_kernel" -- $P/k1s${st}_sq $P/k1s${st}_lds > $P/k1s${st}_counters.json; done
cat $P/k2_counters.json $P/k1s0_counters.json
find $P -name "*_kernel_trace.csv" -size +2M -delete; du -sh $P
