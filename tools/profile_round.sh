#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/prof_<tag>/ (copy what is judged into profiles/ with
# tools/profile_collect.py): rocprofv3 kernel stats of the timed workload of bench.py in BOTH arithmetic
# modes (fp32 = the headline), their bench lines, HBM bytes per kernel from separate FETCH_SIZE / WRITE_SIZE
# PMC passes (both modes), and the same three passes for the standalone projector leg (k_project_deep).
# Usage (on the GPU box): bash tools/profile_round.sh <tag>
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16x3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -- python3 $R/bench.py --steps 20 --warmup 3 --only-timed --precision $mode > $O/bench_profiled_$mode.log 2>&1 || { echo "stats run ($mode) failed"; tail -5 $O/bench_profiled_$mode.log; exit 1; }
  echo "stats $mode done"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$mode -- python3 $R/bench.py --only-timed --steps 3 --warmup 1 --precision $mode > $O/pmc_fetch_$mode.log 2>&1 || { echo "fetch pass ($mode) failed"; tail -5 $O/pmc_fetch_$mode.log; exit 1; }
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$mode -- python3 $R/bench.py --only-timed --steps 3 --warmup 1 --precision $mode > $O/pmc_write_$mode.log 2>&1 || { echo "write pass ($mode) failed"; tail -5 $O/pmc_write_$mode.log; exit 1; }
  echo "pmc $mode done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_proj -- python3 $R/bench.py --leg projector_standalone --steps 20 > $O/bench_profiled_proj.log 2>&1 || { echo "projector stats failed"; tail -5 $O/bench_profiled_proj.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_proj -- python3 $R/bench.py --leg projector_standalone --steps 6 > $O/pmc_fetch_proj.log 2>&1 || { echo "projector fetch pass failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_proj -- python3 $R/bench.py --leg projector_standalone --steps 6 > $O/pmc_write_proj.log 2>&1 || { echo "projector write pass failed"; exit 1; }
echo profiled
