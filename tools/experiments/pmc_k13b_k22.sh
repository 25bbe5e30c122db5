cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; P=gpurun_out/r03d2; mkdir -p $P
for site in k13qkv0 k13fc1s0 k13qkv2 k22; do
  pat=ws_linear_split_kernel; [ $site = k22 ] && pat=ffn_split_kernel
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/${site}_sq -- python3 tools/run_kernel.py $site 12 > $P/${site}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $P/${site}_lds -- python3 tools/run_kernel.py $site 12 > $P/${site}_lds.log 2>&1
  python3 tools/pmc_agg.py --kernels "${site}=${pat}" -- $P/${site}_sq $P/${site}_lds > $P/${site}_counters.json
  tail -1 $P/${site}_sq.log
done
find $P -name "*.db" -delete; find $P -name "*kernel_trace.csv" -size +4M -delete
