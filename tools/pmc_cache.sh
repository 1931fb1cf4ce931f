#!/bin/bash
# L1 / L2 hit rates per kernel of the bench's timed workload (one PMC pass): bash tools/pmc_cache.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_cache
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum \
  --output-format csv -d $R/gpurun_out/pmc_cache -- python3 $R/bench.py ${@:---only-timed --steps 2 --warmup 1 --precision fp32} > $R/gpurun_out/pmc_cache.log 2>&1 || { echo "pass failed"; tail -5 $R/gpurun_out/pmc_cache.log; exit 1; }
cd $R && python3 - <<'PY'
import csv, glob, collections
f = max(glob.glob("gpurun_out/pmc_cache/*/*counter_collection.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-34s %5s %12s %12s %8s %12s %12s %8s %10s" % ("kernel", "n", "L1 acc", "L1->L2 rd", "L1 hit", "L2 hit", "L2 miss", "L2 hit%", "EA rd MB"))
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("TCC_MISS_sum", [0]))):
    n = len(c["TCC_HIT_sum"]); m = lambda x: sum(c[x]) / max(1, len(c[x]))
    acc, rd, hit, miss, ea = m("TCP_TOTAL_CACHE_ACCESSES_sum"), m("TCP_TCC_READ_REQ_sum"), m("TCC_HIT_sum"), m("TCC_MISS_sum"), m("TCC_EA0_RDREQ_sum")
    if hit + miss < 1e4: continue
    print("%-34s %5d %12.3g %12.3g %8.2f %12.3g %12.3g %8.2f %10.1f" % (k, n, acc, rd, 1 - rd / max(acc, 1), hit, miss, hit / max(1, hit + miss), ea * 128 / 1e6))
PY
