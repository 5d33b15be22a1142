#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_c2.py tests/test_gpu_fuzz.py tests/test_gpu_callers.py -x -q 2>&1 | tail -2
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']; print('$1','ms',d['ms_per_step'],'screen_ms',d['roofline'].get('all_screen_kernels_ms_per_step'),'cand',e['candidates_per_query_per_step'],'resc',e['rescored_per_query_per_step'],'retry',e['retry_queries'],'fb',e['fallback_queries'],'kernel',d['roofline']['kernel'])"; }
for rows in 10000000 1250000; do for a in "" "--k 100" "--k 24" "--k 1"; do
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-extras $a 2>/dev/null | tail -1 | line "$rows $a"
done; done
echo "--- C2 (anisotropic, ip, 2M rows)"
for a in "--k 100 --screen auto" "--k 100 --screen i8" "--k 10 --screen auto" "--k 10 --screen i8" "--k 10 --screen bf16" "--k 100 --screen auto --metric cosine"; do
python bench.py --data anisotropic --metric ip --rows 2000000 --steps 10 --warmup 2 --no-cpu-baseline --no-extras $a 2>/dev/null | tail -1 | line "$a"
done
