#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/round_evidence.sh
bash tools/collect_l1path.sh > gpurun_out/ev/l1path.log 2>&1; tail -3 gpurun_out/ev/l1path.log
ROWS=10000000 BENCH_ARGS="--no-extras" bash tools/step_timeline.sh > gpurun_out/ev/timeline_10m.txt 2>&1; tail -2 gpurun_out/ev/timeline_10m.txt
timeout 400 python tools/fuzz_parity.py --seconds 300 2>&1 | tail -1 | cut -c1-300
