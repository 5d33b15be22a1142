#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1','ms',d['ms_per_step'],'screen_ms',d['roofline'].get('all_screen_kernels_ms_per_step'),'big_ms',d['roofline']['kernel_ms_per_step'],'cand',d['extra']['candidates_per_query_per_step'],'resc',d['extra']['rescored_per_query_per_step'],'retry',d['extra']['retry_queries'])"; }
for rows in 10000000 1250000; do for a in "" "--defer-b 0" "--growth 3" "--growth 3 --defer-b 0" "--starter 0 --defer-b 0 --sync-steps"; do
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-extras $a 2>/dev/null | tail -1 | line "$rows $a"
done; done
bash tools/shard_sizes.sh
echo "--- C2 (anisotropic, ip, k=100, 2M rows)"
c2line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']; print('$1','ms',d['ms_per_step'],'screen_ms',d['roofline'].get('all_screen_kernels_ms_per_step'),'cand',e['candidates_per_query_per_step'],'resc',e['rescored_per_query_per_step'],'retry',e['retry_queries'],'fb',e['fallback_queries'],'kernel',d['roofline']['kernel'])"; }
for a in "--screen auto" "--screen bf16" "--screen i8" "--screen auto --k 10" "--screen i8 --k 10" "--screen bf16 --k 10"; do
python bench.py --data anisotropic --metric ip --k 100 --rows 2000000 --steps 10 --warmup 2 --no-cpu-baseline --no-extras $a 2>/dev/null | tail -1 | c2line "$a"
done
