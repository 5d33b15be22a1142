#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# third form: 202024 default ; 202280 = early release ; 202792 = early release + the two waves of a SIMD issue DMA in alternate micro-steps
echo "== candidate sets equal?"; VARIANTS=4436,202792 timeout 120 tools/bin/screen_bench 1048576 1024 768 2 2>&1 | grep "candidate set"
for r in 1 2; do
VARIANTS=202024,202280,202792 ROUNDS=11 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -3
done
