#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=202024,202152,202280,202408
echo "== candidate sets equal?"; VARIANTS=4436,$V timeout 120 tools/bin/screen_bench 1048576 1024 768 2 2>&1 | grep "candidate set"
# LDS-DMA piece schedules kc_sched(id): 202024 + 128 id
for r in 1 2; do
VARIANTS=$V ROUNDS=9 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -8
done
