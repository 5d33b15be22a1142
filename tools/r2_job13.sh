#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j13; mkdir -p $OUT; rm -f $OUT/*.json
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_c2.py tests/test_gpu_maxsim.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/a_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j13/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], 'rest', round(d['ms_per_step']-r['all_screen_kernels_ms_per_step'],3))
PY
bash tools/step_timeline.sh 2>&1 | grep "k_prune<64" | head -8
