#!/bin/bash
# chunk schedule at N = 10 M with the third-form screen: first chunk, growth, small-chunk kernel switch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$*', 'ms',d['ms_per_step'],'qps',round(d['value']), 'chunks', c.get('chunks_per_pass'), 'cand', c.get('candidates_per_query'), 'resc', c.get('rescored_per_query'), 'retry', c.get('retry_queries'))"; }
run
run --growth 4
run --growth 2
run --chunk0 2048
run --chunk0 4096
run --chunk0 4096 --growth 4
run --small-chunk 4096
run --small-chunk 65536
run
