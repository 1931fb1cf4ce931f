#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/prof_<tag>/ (copy what is judged into profiles/):
#   kernel stats of the default bench run, its bench line, and HBM bytes per kernel from two PMC passes.
# Usage (on the GPU box): bash tools/profile_round.sh <tag>
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --only-timed > $O/bench_profiled.log 2>&1 || { echo "stats run failed"; tail -5 $O/bench_profiled.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --only-timed --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $O/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --only-timed --steps 3 --warmup 1 > $O/pmc_write.log 2>&1 || { echo "write pass failed"; tail -5 $O/pmc_write.log; exit 1; }
timeout -k 10 120 $R/tools/micro/reqbench > $O/reqbench.txt 2>&1 || echo "reqbench failed"
echo profiled
