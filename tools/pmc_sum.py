import csv,glob,collections,sys
kern=sys.argv[1]; tags=sys.argv[2]
for tag in tags:
    fs=glob.glob(f"gpurun_out/pmc_va_{tag}/*/*counter_collection.csv")
    if not fs: print(tag,"missing"); continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list)
    for r in rows:
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(tag,k,"n=%d"%len(v),"mean=%.4g"%(sum(v)/len(v)))
