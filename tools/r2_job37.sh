#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rows in 2000000 10000000; do for scr in auto bf16; do
python bench.py --data anisotropic --k 10 --rows $rows --screen $scr --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; e=d['extra']; print('aniso',$rows,'k 10','$scr','ms',d['ms_per_step'],r['kernel'],'cand',e['candidates_per_query_per_step'],'resc',e['rescored_per_query_per_step'],'fallback',e['fallback_queries'],'loose',e['loose_rows'])"
done; done
