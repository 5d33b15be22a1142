#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== candidate sets equal?"; VARIANTS=4436,204072 timeout 120 tools/bin/screen_bench 1048576 1024 768 2 2>&1 | grep "candidate set"
for r in 1 2; do
VARIANTS=202024,204072 ROUNDS=11 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -2
done
