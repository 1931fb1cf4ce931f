#!/bin/bash
# Memory-side PMC passes over the view-attention microbench (one pass per counter group).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "D:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "E:TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum" \
           "F:TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "G:TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "H:TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_HIT_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "I:TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum"; do
  tag=${grp%%:*}; ctr=${grp#*:}
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_va_$tag -- python3 $R/tools/vabench.py bf16x3 one > $R/gpurun_out/pmc_va_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $R/gpurun_out/pmc_va_$tag.log; }
done
echo done
