#!/bin/bash
# MaxSim evidence (round 3): rocprofv3 --kernel-trace --stats and PMC passes (separate, kernel-trace only) of the MaxSim bench at
# the survey's sizes -> gpurun_out/r3ms/ ; summaries are copied into profiles/r03_maxsim_* by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3ms; mkdir -p $OUT
for shape in text page; do
  docs=${DOCS_TEXT:-1000000}; [ $shape = page ] && docs=${DOCS_PAGE:-100000}
  ARGS="--workload maxsim --tokens $shape --docs $docs --steps 12 --warmup 2 --no-cpu-baseline"
  python bench.py $ARGS --steps 125 > $OUT/line_$shape.json 2> $OUT/line_$shape.err; cut -c1-900 $OUT/line_$shape.json
  rm -rf $OUT/st_$shape $OUT/pmc_$shape
  MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --stats -f csv -d $OUT/st_$shape -o ms -- python bench.py $ARGS > $OUT/st_$shape.log 2>&1
  MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_$shape -o fetch -- python bench.py $ARGS > $OUT/pmc_${shape}_fetch.log 2>&1
  MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d $OUT/pmc_$shape -o sq -- python bench.py $ARGS > $OUT/pmc_${shape}_sq.log 2>&1
  python - "$OUT" "$shape" "$docs" <<'PY'
import collections, csv, glob, json, sys
out_dir, shape, docs = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out_dir}/pmc_{shape}/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "k_maxsim" in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
line = json.loads(open(f"{out_dir}/line_{shape}.json").read().strip().splitlines()[-1])
tu = line["roofline"]["traffic_unit"]
streamed = float(tu.split("and ")[1].split(" streamed")[0])
res = {"shape": shape, "docs": docs, "bench_line_roofline": line["roofline"], "kernels": {}}
for kname, cs in agg.items():
    o = {"launches_profiled": len(next(iter(cs.values())))}
    for c, v in cs.items():
        o[c + "_mean_per_launch"] = sum(v) / len(v)
    if "FETCH_SIZE_mean_per_launch" in o:
        o["hbm_read_bytes_per_launch_corrected"] = o["FETCH_SIZE_mean_per_launch"] * 1024 * 2  # KiB; gfx950: 128-B requests counted as 64 B
    if "SQ_VALU_MFMA_BUSY_CYCLES_mean_per_launch" in o and "GRBM_GUI_ACTIVE_mean_per_launch" in o:
        o["mfma_busy_fraction_at_actual_clock"] = o["SQ_VALU_MFMA_BUSY_CYCLES_mean_per_launch"] / (o["GRBM_GUI_ACTIVE_mean_per_launch"] / 8 * 1024)
    res["kernels"][kname] = o
    if "k_maxsim16" in kname and "hbm_read_bytes_per_launch_corrected" in o:
        res["hbm_read_bytes_per_streamed_byte"] = o["hbm_read_bytes_per_launch_corrected"] / streamed
        res["screen_kernel"] = kname
json.dump(res, open(f"{out_dir}/r03_maxsim_traffic_{shape}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "bench_line_roofline"}, indent=1)[:1800])
PY
  cp $OUT/st_$shape/ms_kernel_stats.csv $OUT/r03_maxsim_${shape}_kernel_stats.csv 2>/dev/null
done
