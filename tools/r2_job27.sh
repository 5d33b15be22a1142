#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROWS=2000000 BENCH_ARGS="--data anisotropic --metric ip --k 100 --no-extras" bash tools/step_timeline.sh > gpurun_out/c2_timeline.txt 2>&1
python - <<'PY'
import re,collections
agg=collections.defaultdict(lambda:[0,0.0])
for l in open('gpurun_out/c2_timeline.txt'):
    m=re.match(r'\s*([\d.]+) us\s+gap\s+([\d.]+)\s+dur\s+([\d.]+)\s+(.*?)\s+grid',l)
    if m:
        n=m.group(4)[:60]; agg[n][0]+=1; agg[n][1]+=float(m.group(3))
for n,(c,t) in sorted(agg.items(), key=lambda x:-x[1][1])[:14]: print(f"{t:9.1f} us {c:4d}  {n}")
PY
tail -1 gpurun_out/c2_timeline.txt
