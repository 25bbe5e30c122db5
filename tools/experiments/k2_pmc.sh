cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; P=gpurun_out/k2pmc; mkdir -p $P
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msda" 2>&1 | tail -1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum -d $P/k2_tcp -- python3 tools/k2_probe.py 12 plain > $P/k2_tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $P/k2_tcc -- python3 tools/k2_probe.py 12 plain > $P/k2_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $P/k2_sq -- python3 tools/k2_probe.py 12 plain > $P/k2_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM -d $P/k2_sq2 -- python3 tools/k2_probe.py 12 plain > $P/k2_sq2.log 2>&1
python3 tools/pmc_agg.py --kernels "k2_fused=msda_fused_tiles_kernel" -- $P/k2_tcp $P/k2_tcc $P/k2_sq $P/k2_sq2 > $P/k2_counters.json
python3 tools/k2_probe.py 50 plain | tail -1
find $P -name "*.db" -delete
