#!/bin/bash
# kernel timeline of one 1024-query pass at N = 10 M and at the 8-way shard size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j18
ROWS=10000000 BENCH_ARGS="--no-extras" bash tools/step_timeline.sh > gpurun_out/j18/timeline_10m.txt 2>&1; tail -45 gpurun_out/j18/timeline_10m.txt
ROWS=1250000 BENCH_ARGS="--no-extras" bash tools/step_timeline.sh > gpurun_out/j18/timeline_1250k.txt 2>&1; tail -3 gpurun_out/j18/timeline_1250k.txt
