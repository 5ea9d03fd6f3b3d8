# SQ counters of the convolution kernels (two rocprofv3 --pmc passes, never combined with tracing).
# (A TA_* / TD_* pass and a TCP_* pass of the same command did not finish within 15 minutes on this
# pool and are not collected.)  Usage, through gpurun:  bash scripts/pmc_conv_probe.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcx; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/scripts/pool_fused_ab.py --iters 2 --layers 1_2,3_3 > /dev/null 2>&1
done
cd $R
python3 scripts/pmc_summary.py $O --only kernel --out $O/summary.csv > /dev/null 2>&1
ls $O; rm -rf $O/p1 $O/p2
