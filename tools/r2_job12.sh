#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== candidate sets equal?"; VARIANTS=4436,206120,210216 timeout 120 tools/bin/screen_bench 1048576 1024 768 2 2>&1 | grep "candidate set"
# 202024 default | 206120 six pieces behind the hand-over (2 2 2 | 2 1) | 210216 s_setprio 1 around the MFMAs
for r in 1 2; do
VARIANTS=202024,206120,210216 ROUNDS=11 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -3
done
