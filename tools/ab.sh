#!/bin/bash
# A/B of two builds of the library on one box: tools/ab.sh libA.so libB.so [steps]  (alternating runs)
A=$1; B=$2; steps=${3:-60}
for rep in 1 2; do
  for lib in $A $B; do
    GPN_LIB=$PWD/$lib timeout -k 10 200 python bench.py --only-timed --precision ${PREC:-bf16x3} --steps $steps --warmup 5 > /tmp/ab.log 2>&1 || { tail -5 /tmp/ab.log; exit 1; }
    python - $lib <<'PY'
import json,sys
l=[x for x in open("/tmp/ab.log") if x.startswith("{")][-1]; d=json.loads(l)
print("%-32s rays/s %.0f  ms/step %.3f  %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: v for k, v in d["stage_ms_one_step"].items() if v > 0.2}), flush=True)
PY
  done
done
