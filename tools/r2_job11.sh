#!/bin/bash
# interleaved A/B on one box: previous commit (ab_old/) vs working tree; bench lines + parked kernel + hit sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j11; mkdir -p $OUT; rm -f $OUT/*.json
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do
  (cd ab_old && timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > ../$OUT/old_$r.json)
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/new_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j11/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], d['extra']['candidates_per_query_per_step'])
PY
echo "== parked + sweep, new kernel =="; VARIANTS=4436 SWEEP=1 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | grep -v threshold
