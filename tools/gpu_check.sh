#!/bin/bash
# GPU check used during development: parity tests, then a short bench.  Usage: tools/gpu_check.sh [tag]
tag=${1:-run}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/test_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --cpu-rays 0 --steps 20 --warmup 3 > gpurun_out/bench_$tag.log 2>&1 || { tail -5 gpurun_out/bench_$tag.log; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$tag.log") if x.startswith("{")][-1]
d=json.loads(l)
print("rays/s %.0f  ms/step %.3f  view_attn launch ms %.4f  frac %.3f  fp32 %.0f"%(d["value"],d["ms_per_step"],d["roofline"]["mean_launch_ms"],d["roofline"]["frac"],d["fp32_exact"]["value"]))
print(d["stage_ms_one_step"])
PY
