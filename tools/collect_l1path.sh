#!/bin/bash
# On the GPU box: counters of the vector-memory (TA/TD/TCP) path and of the issue mix of the screen kernel over the
# default bench command -- separate kernel-trace-only PMC passes; raw CSVs -> gpurun_out/l1path/, summary printed.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/l1path; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $@"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT -o $name -- python bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed: $(tail -2 $OUT/$name.log)"; }
pass ta1 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass ta3 TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
pass td1 TD_TD_BUSY_sum TD_TC_STALL_sum
pass tcp1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp3 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
pass sq1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAVES
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/l1path/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "screen256" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"n": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in sorted(cs.items())} for k, cs in agg.items()}
json.dump(out, open("gpurun_out/l1path/summary.json", "w"), indent=1)
for k, cs in out.items():
    print("==", k)
    for c, v in cs.items(): print(f"   {c:40s} n={v['n']:3d} mean={v['mean_per_launch']:.5g}")
PY
