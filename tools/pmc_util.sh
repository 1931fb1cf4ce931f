#!/bin/bash
# MFMA / issue utilisation per kernel of the bench's timed workload, one SQ pass + GRBM_GUI_ACTIVE per mode:
#   bash tools/pmc_util.sh            (on the GPU box; summary: python tools/pmc_util.py -> profiles/r03_pmc_util.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in fp32 bf16x3 bf16; do
  rm -rf $R/gpurun_out/pmc_util_$mode
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    --output-format csv -d $R/gpurun_out/pmc_util_$mode -- python3 $R/bench.py --only-timed --steps 2 --warmup 1 --precision $mode > $R/gpurun_out/pmc_util_$mode.log 2>&1 || { echo "pass $mode failed"; tail -5 $R/gpurun_out/pmc_util_$mode.log; exit 1; }
  echo "pass $mode done"
done
