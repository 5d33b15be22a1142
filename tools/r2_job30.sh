#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -1
for r in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; done
ROWS=10000000 BENCH_ARGS="--no-extras" bash tools/step_timeline.sh > gpurun_out/tl30.txt 2>&1; grep "k_prune<64" gpurun_out/tl30.txt | awk '{s+=$6; printf "%s ", $6} END{print "sum",s}'
