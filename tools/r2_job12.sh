#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 4436 every K-step | 8532 never | 12628 only at the last K-step | 20820 every K-step, wait pinned after the last MFMA | 29012 both
for r in 1 2; do
VARIANTS=4436,8532,12628,20820,29012 ROUNDS=15 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -5
done
