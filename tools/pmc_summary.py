"""Dev helper: summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, sys, glob, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "screen" not in k: continue
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
