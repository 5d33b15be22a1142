#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== through the gather + merge path (world 1)"; bash tools/shard_sizes.sh 2>&1 | tail -4
echo "== plain search"; for rows in 10000000 5000000 2500000 1250000; do python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rows',$rows,'ms',d['ms_per_step'],'qps',d['value'], d['extra']['candidates_per_query_per_step'])"; done
