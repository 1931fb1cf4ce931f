#!/bin/bash
# Rehearsal of bench.py's multi-rank control flow on a ONE-GPU box: N ranks share cuda:0 and talk over
# gloo (GPN_BENCH_REHEARSE=1).  Checks that every rank reaches the end (no rank-local collective) and
# that rank 0 prints one JSON line; the throughput printed is meaningless (the tiles are gathered as host copies
# over gloo).  Last run: gpurun_out/rehearse2.log, 2 ranks, rc 0, one line.  Usage: tools/rehearse_ranks.sh [N]
N=${1:-2}
GPN_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 3 --warmup 1
