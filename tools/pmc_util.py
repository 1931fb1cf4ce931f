"""Summary of tools/pmc_util.sh: per kernel (mean over its launches) the launch time, the effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / time), the share of the SIMD cycles in which the MFMA pipe is busy
(SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)) and the split of the wave cycles into issuing /
stalled at issue / waiting (quad-cycle counters, MI355X_MICROARCH.md).  Writes profiles/r03_pmc_util.txt."""
import collections
import csv
import glob
import os

out = ["# rocprofv3 --pmc pass over `bench.py --only-timed --steps 2 --warmup 1 --precision <mode>` (tools/pmc_util.sh)",
       "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / launch time",
       "# (reads high on launches well under 0.3 ms); wave-cycle split: issuing / stalled at issue / waiting (s_waitcnt, barrier)"]
for mode in ("fp32", "bf16x3", "bf16"):
    fs = glob.glob("gpurun_out/pmc_util_%s/*/*counter_collection.csv" % mode)
    if not fs:
        continue
    f = max(fs, key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("k_"):
            continue
        d = per[name]
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            d["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out.append("")
    out.append("## %s" % mode)
    out.append("%-28s %6s %9s %8s %9s %8s %8s %8s %12s" % ("kernel", "calls", "us/launch", "GHz", "mfma_busy", "issuing", "stalled", "waiting", "MFMA insts"))
    rows = []
    for name, d in per.items():
        m = {k: sum(v) / len(v) for k, v in d.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        wave = max(1.0, m["SQ_WAVE_CYCLES"])
        rows.append((m["ns"] * len(d["ns"]), "%-28s %6d %9.1f %8.2f %9.2f %8.2f %8.2f %8.2f %12.3g" % (
            name[:28], len(d["ns"]), m["ns"] / 1e3, cyc / m["ns"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
            m["SQ_ACTIVE_INST_ANY"] / wave, m["SQ_WAIT_INST_ANY"] / wave, m["SQ_WAIT_ANY"] / wave, m["SQ_INSTS_MFMA"])))
    out += [r for _, r in sorted(rows, reverse=True)]
open("profiles/r03_pmc_util.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
