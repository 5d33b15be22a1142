#!/bin/bash
# address-translation and L2-latency counters of k_prune over the default bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j29; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT -o $name -- python bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed: $(tail -2 $OUT/$name.log)"; }
pass u1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass u2 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass u3 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
pass u4 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass u5 TCC_HIT_sum TCC_MISS_sum
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/j29/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_prune<64" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print("==", k)
    for c, v in sorted(cs.items()): print(f"   {c:44s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
