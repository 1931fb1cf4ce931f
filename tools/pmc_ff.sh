#!/bin/bash
# PMC passes over the feed-forward microbench (tools/ffbench.py) (one pass per counter group).
cd /tmp && export TMPDIR=/tmp FF_NS=262144
R=$GRAFT_REPO_ROOT
for grp in "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "B:SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "C:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  tag=${grp%%:*}; ctr=${grp#*:}
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_ff_$tag -- python3 $R/tools/ffbench.py $R/gp-nerf_amd/csrc/libgpnerf_hip.so > $R/gpurun_out/pmc_ff_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $R/gpurun_out/pmc_ff_$tag.log; exit 1; }
done
echo done
