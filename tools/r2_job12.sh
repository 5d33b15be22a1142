#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j12; mkdir -p $OUT; rm -f $OUT/*.json
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_c2.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
echo "== old"; VARIANTS=4436 ROUNDS=15 timeout 300 ab_old/tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -1
echo "== new"; VARIANTS=4436 ROUNDS=15 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -1
done
echo "== new sweep"; VARIANTS=4436 SWEEP=1 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | grep -v threshold
for r in 1 2; do
  (cd ab_old && timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > ../$OUT/old_$r.json)
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/new_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j12/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], d['extra']['candidates_per_query_per_step'], r['all_screen_launches'])
PY
timeout 200 python tools/fuzz_parity.py --seconds 90 2>&1 | tail -1 | cut -c1-200
