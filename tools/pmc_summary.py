"""Dev helper: summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch) as JSON.
usage: python tools/pmc_summary.py <dir> [kernel-name substring, default "screen"]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "screen"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if want in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    o = {"launches_profiled": len(next(iter(cs.values())))}
    for c, v in sorted(cs.items()):
        o[c + "_mean_per_launch"] = sum(v) / len(v)
    if "SQ_VALU_MFMA_BUSY_CYCLES_mean_per_launch" in o and "GRBM_GUI_ACTIVE_mean_per_launch" in o:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE over 8 XCDs
        o["mfma_busy_fraction_at_actual_clock"] = o["SQ_VALU_MFMA_BUSY_CYCLES_mean_per_launch"] / (o["GRBM_GUI_ACTIVE_mean_per_launch"] / 8 * 1024)
    out[k] = o
print(json.dumps(out, indent=1))
