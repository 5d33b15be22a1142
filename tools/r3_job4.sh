#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for db in 1 0 1 0; do
echo "== defer-b $db"
STEPS=400 BENCH_ARGS="--defer-b $db" bash tools/power_probe.sh 2>&1 | grep -E '"ms_per_step"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/'
python - <<'PY'
import re
rows=[]
for ln in open("gpurun_out/power/samples_bench.txt"):
    m=re.search(r"sclk clock level: \S+ \((\d+)Mhz\).*Power \(W\): ([\d.]+)", ln)
    if m: rows.append((int(m.group(1)), float(m.group(2))))
load=[r for r in rows if r[1]>1000]
if load:
    import statistics as st
    print("  load samples", len(load), "sclk median", st.median(r[0] for r in load), "min", min(r[0] for r in load), "max", max(r[0] for r in load), "| power median", st.median(r[1] for r in load), "max", max(r[1] for r in load))
PY
done
