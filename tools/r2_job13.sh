#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j13; mkdir -p $OUT; rm -f $OUT/*.json
echo "== old"; VARIANTS=4436 ROUNDS=15 timeout 300 ab_old/tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -1
echo "== new: 4436 = reload every K-step, 8532 = no reload"; VARIANTS=4436,8532 ROUNDS=15 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -2
echo "== old"; VARIANTS=4436 ROUNDS=15 timeout 300 ab_old/tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -1
echo "== new"; VARIANTS=4436,8532 ROUNDS=15 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -2
for r in 1 2; do
  (cd ab_old && timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > ../$OUT/old_$r.json)
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/new_$r.json
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --small-chunk 70000 2>/dev/null | tail -1 > $OUT/sc70k_$r.json
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --small-chunk 270000 2>/dev/null | tail -1 > $OUT/sc270k_$r.json
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --growth 4 2>/dev/null | tail -1 > $OUT/g4_$r.json
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --growth 5 2>/dev/null | tail -1 > $OUT/g5_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j13/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], d['extra']['candidates_per_query_per_step'], r['all_screen_launches'])
PY
