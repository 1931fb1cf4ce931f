#!/bin/bash
# One bench leg per library given: tools/legs.sh <leg> libA.so libB.so ...
leg=$1; shift
for lib in "$@"; do
  GPN_LIB=$PWD/$lib timeout -k 10 200 python bench.py --leg $leg --steps ${STEPS:-20} > /tmp/leg.log 2>&1 || { tail -5 /tmp/leg.log; exit 1; }
  python - $lib <<'PY'
import json,sys
l=[x for x in open("/tmp/leg.log") if x.startswith("{")][-1]; d=json.loads(l)
if "kernel_ms" in d:
    print("%-28s ms/call %.3f  kernels %s  frac %.3f" % (sys.argv[1], d["ms_per_call"], {k: round(v, 3) for k, v in d["kernel_ms"].items()}, d["roofline"]["frac"]), flush=True)
else:
    print("%-28s %s" % (sys.argv[1], {m: round(d[m]["ms_per_step"], 3) for m in d if isinstance(d[m], dict) and "ms_per_step" in d[m]}), flush=True)
PY
done
