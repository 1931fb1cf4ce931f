#!/bin/bash
# SQ / TCP PMC passes over a short bench run (all kernels); summarise with tools/pmc_sum.py <kernel> AG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "B:SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "G:TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  tag=${grp%%:*}; ctr=${grp#*:}
  rm -rf $R/gpurun_out/pmc_va_$tag
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_va_$tag -- python3 $R/bench.py --cpu-rays 0 --steps 2 --warmup 1 > $R/gpurun_out/pmc_va_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $R/gpurun_out/pmc_va_$tag.log; }
done
echo done
