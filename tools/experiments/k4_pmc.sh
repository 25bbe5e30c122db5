cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; P=gpurun_out/k4pmc; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/tr -- python3 tools/run_kernel.py dyn 30 > $P/tr.log 2>&1
grep dyn_mask $P/tr/*/*kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $P/sq -- python3 tools/run_kernel.py dyn 12 > $P/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_WR -d $P/sq2 -- python3 tools/run_kernel.py dyn 12 > $P/sq2.log 2>&1
python3 tools/pmc_agg.py --kernels "k4=dyn_mask_kernel" -- $P/sq $P/sq2 > $P/k4_counters.json
tail -3 $P/sq2.log
find $P -name "*.db" -delete
