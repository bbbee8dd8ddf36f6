cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc5 -o p5 -- $B > gpurun_out/pmc5.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc6 -o p6 -- $B > gpurun_out/pmc6.log 2>&1
echo rc=$?
