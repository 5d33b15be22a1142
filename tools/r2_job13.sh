#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j13; mkdir -p $OUT; rm -f $OUT/*.json
for r in 1 2 3; do
  for a in 0 -1; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --round-a $a 2>/dev/null | tail -1 > $OUT/ra${a}_$r.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j13/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], r['all_screen_kernels_ms_per_step'], 'rest', round(d['ms_per_step']-r['all_screen_kernels_ms_per_step'],3), d['extra']['candidates_per_query_per_step'], d['extra']['rescored_per_query_per_step'])
PY
