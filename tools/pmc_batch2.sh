L="--cin 196 --cout 196 --k 3 --act leaky --iters 8"
for mode in 0 2; do
export GIM_IGEMM_RING3=$mode
echo "=== RING3 mode $mode"
timeout 100 python tools/microbench_conv.py $L 2>&1 | grep -v amdgpu
tools/pmc_run.sh h$mode SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- $L 2>&1 | grep -v amdgpu
tools/pmc_run.sh i$mode SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU -- $L 2>&1 | grep -v amdgpu
tools/pmc_run.sh j$mode TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -- $L 2>&1 | grep -v amdgpu
done
