#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for pers in 1 0; do for shape in page text; do docs=100000; [ $shape = text ] && docs=1000000
MI355DR_MAXSIM_PERSISTENT=$pers python bench.py --workload maxsim --tokens $shape --docs $docs --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('persistent $pers $shape', 'ms/step', d['ms_per_step'], 'screen ms', r['avg_launch_ms'], 'streamed GB/s', r['streamed_GBps'])"
done; done; done
