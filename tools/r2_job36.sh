#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 25 200 400; do for scr in auto i8; do
python bench.py --k $k --screen $scr --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; e=d['extra']; print('k',$k,'$scr','ms',d['ms_per_step'],'qps',d['value'],r['kernel'],'launches',r['all_screen_launches'],'cand',e['candidates_per_query_per_step'],'resc',e['rescored_per_query_per_step'],'fallback',e['fallback_queries'])"
done; done
