#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_maxsim.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --workload maxsim --docs 100000 --steps 25 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('text 100k', d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('cpu_baseline',{}).get('parity_on_sample'))"
timeout 300 python bench.py --workload maxsim --docs 20000 --tokens page --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('page 20k', d['value'], d['ms_per_step'])"
timeout 200 python tools/fuzz_parity.py --seconds 120 --only maxsim 2>&1 | tail -1 | cut -c1-200
