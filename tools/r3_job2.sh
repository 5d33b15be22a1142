#!/bin/bash
# deferred round B: parity, then A/B at both sizes + fuzz
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_c2.py tests/test_gpu_callers.py -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1','ms',d['ms_per_step'],'screen_ms',d['roofline'].get('all_screen_kernels_ms_per_step'),'cand',d['extra']['candidates_per_query_per_step'],'resc',d['extra']['rescored_per_query_per_step'],'fb',d['extra']['fallback_queries'])"; }
for rep in 1 2; do for rows in 10000000 1250000; do
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | line "rows $rows default  "
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-extras --defer-b 0 2>/dev/null | tail -1 | line "rows $rows defer-b 0"
done; done 2>&1 | tee $OUT/ab.log
timeout 600 python tools/fuzz_parity.py --seconds 240 --seed 31337 2>&1 | tail -4
ROWS=1250000 BENCH_ARGS=--no-extras bash tools/step_timeline.sh > $OUT/timeline_1250k.txt 2>&1; tail -22 $OUT/timeline_1250k.txt
