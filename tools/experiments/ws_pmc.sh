cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; P=gpurun_out/wspmc; mkdir -p $P
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/a -- python3 tools/ws_probe.py > $P/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES -d $P/b -- python3 tools/ws_probe.py > $P/b.log 2>&1
for k in "ws_linear_kernel<96, 0, true, false" "ws_linear_kernel<96, 2, true, false" "ws_linear_kernel<384, 0, false, true" "ws_linear_kernel<96, 0, false, true"; do
python3 tools/pmc_agg.py --kernels "k=$k" -- $P/a $P/b | python3 -c "
import json,sys; d=json.load(sys.stdin)
print('$k')
for kk,v in d.items():
    if isinstance(v,dict):
        for a,b in v.items():
            if isinstance(b,dict): print('   ',a, round(b['mean_per_launch']), b['launches'])
            else: print('   ',a,b)
"
done
find $P -name "*.db" -delete
