#!/bin/bash
# One bench run per library given (stage table of each): tools/abn.sh libA.so libB.so ...   (PREC=fp32|bf16x3)
for lib in "$@"; do
  GPN_LIB=$PWD/$lib timeout -k 10 200 python bench.py --only-timed --precision ${PREC:-bf16x3} --steps ${STEPS:-20} --warmup 3 > /tmp/ab.log 2>&1 || { tail -5 /tmp/ab.log; exit 1; }
  python - $lib <<'PY'
import json,sys
l=[x for x in open("/tmp/ab.log") if x.startswith("{")][-1]; d=json.loads(l)
print("%-44s ms/step %.3f  %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], {k: v for k, v in d["stage_ms_one_step"].items() if v > 0.1}), flush=True)
PY
done
