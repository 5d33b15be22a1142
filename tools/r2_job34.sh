#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 1 8 32 64; do
for g in 0 3; do
python bench.py --block $B --growth $g --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; e=d['extra']; print('B',$B,'growth',$g,'ms',d['ms_per_step'],'qps',d['value'],'screen_ms',r.get('all_screen_kernels_ms_per_step'),'launches',r.get('all_screen_launches'),'cand',e['candidates_per_query_per_step'],'fallback',e['fallback_queries'])"
done; done
timeout 600 python -m pytest tests/test_gpu_search.py -m gpu -x -q 2>&1 | tail -1
