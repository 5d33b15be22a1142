#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3c; mkdir -p $OUT
for db in 1 0 1 0; do
ROWS=10000000 STEPS=12 BENCH_ARGS="--no-extras --defer-b $db" bash tools/step_timeline.sh > $OUT/timeline_10m_defer$db.txt 2>&1; echo "== defer-b $db"; grep -E "k_screen|k_prune|span" $OUT/timeline_10m_defer$db.txt | awk '{print $7, $8, $9}' | tr '\n' ' '; echo
done
