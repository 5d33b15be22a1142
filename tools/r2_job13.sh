#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j13; mkdir -p $OUT; rm -f $OUT/*.json
for r in 1 2; do
  for g in 3 4 5 6; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --growth $g 2>/dev/null | tail -1 > $OUT/g${g}_$r.json
  done
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --growth 4 --small-chunk 70000 2>/dev/null | tail -1 > $OUT/g4sc70k_$r.json
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --growth 4 --chunk0 2048 2>/dev/null | tail -1 > $OUT/g4c2048_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j13/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], d['extra']['candidates_per_query_per_step'], d['extra']['rescored_per_query_per_step'], r['all_screen_launches'], d['extra']['fallback_queries'])
PY
