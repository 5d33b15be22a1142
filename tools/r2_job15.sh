#!/bin/bash
# fourth form (k_screen256d): parked timing vs forms two and three, correctness with MI355DR_SCREEN_FORM=3, bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j15; mkdir -p $OUT
echo "== candidate sets equal?"; VARIANTS=4436,301000 timeout 120 tools/bin/screen_bench 1048576 1024 768 2 2>&1 | tail -4
echo "== bf16"; VARIANTS=600,300000 timeout 120 tools/bin/screen_bench 2097152 1024 768 2 2>&1 | tail -5
for r in 1 2; do VARIANTS=4436,202024,301000,301016 ROUNDS=11 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -4; done
echo "== sweep d"; VARIANTS=301000 SWEEP=1 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | grep -v threshold
export MI355DR_SCREEN_FORM=3
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_c2.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/d.json
unset MI355DR_SCREEN_FORM
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/b.json
MI355DR_SCREEN_FORM=2 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/c.json
python - <<'PY'
import json
for f in ('b','c','d'):
    d=json.load(open(f'gpurun_out/j15/{f}.json')); r=d['roofline']
    print(f, d['ms_per_step'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], d['extra']['candidates_per_query_per_step'])
PY
MI355DR_SCREEN_FORM=3 timeout 200 python tools/fuzz_parity.py --seconds 120 2>&1 | tail -1 | cut -c1-200
