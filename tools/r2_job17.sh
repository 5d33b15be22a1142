#!/bin/bash
# tables of DESIGN sections 4.1 / 5 / 6 with the third-form default: shard sizes, block sizes, k, C2, MaxSim
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j17; mkdir -p $OUT
echo "== shard sizes"; bash tools/shard_sizes.sh 2>&1 | tail -4
echo "== shard sizes, plain search"; for rows in 5000000 2500000 1250000; do python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rows',$rows,'ms',d['ms_per_step'],'qps',d['value'], d['extra']['candidates_per_query_per_step'])"; done
echo "== block sweep"; sed -i 's/--no-cpu-baseline 2>/--no-cpu-baseline --no-extras 2>/' tools/block_sweep.sh; bash tools/block_sweep.sh 2>&1 | tail -6
echo "== k sweep"; for k in 1 10 24 100; do python bench.py --k $k --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('k',$k,'ms',d['ms_per_step'],'qps',d['value'],r['kernel'],'cand',d['extra']['candidates_per_query_per_step'],'resc',d['extra']['rescored_per_query_per_step'])"; done
echo "== ip"; python bench.py --metric ip --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
echo "== C2"; python bench.py --data anisotropic --metric ip --k 100 --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/c2.json; python -c "
import json; d=json.load(open('$OUT/c2.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['extra'])"
