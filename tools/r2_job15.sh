#!/bin/bash
# third form (k_screen256c): parked timing vs the second form, then correctness with MI355DR_SCREEN_FORM=2, then the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j15; mkdir -p $OUT
for r in 1 2; do VARIANTS=4436,202024,202028 ROUNDS=11 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -4; done
echo "== sweep c"; VARIANTS=202024 SWEEP=1 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | grep -v threshold
echo "== candidate sets equal?"; VARIANTS=4436,202024 timeout 300 tools/bin/screen_bench 1048576 1024 768 2 2>&1 | tail -3
echo "== bf16"; VARIANTS=600,200000 timeout 300 tools/bin/screen_bench 4194304 1024 768 3 2>&1 | tail -6
export MI355DR_SCREEN_FORM=2
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_c2.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/c.json
unset MI355DR_SCREEN_FORM
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/b.json
python - <<'PY'
import json
for f in ('b','c'):
    d=json.load(open(f'gpurun_out/j15/{f}.json')); r=d['roofline']
    print(f, d['ms_per_step'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], d['extra']['candidates_per_query_per_step'])
PY
MI355DR_SCREEN_FORM=2 timeout 200 python tools/fuzz_parity.py --seconds 120 2>&1 | tail -1 | cut -c1-200
