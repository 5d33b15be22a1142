#!/bin/bash
# cache policy of the LDS-DMA pieces (k_screen256c): 202024 default | 206120 corpus nt | 210216 corpus sc1 | 214312 corpus sc0 sc1
# | 218408 queries nt | 222504 both nt | 234792 queries sc1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=202024,206120,210216,214312,218408,222504,234792
echo "== candidate sets equal?"; VARIANTS=4436,$V timeout 120 tools/bin/screen_bench 1048576 1024 768 2 2>&1 | grep "candidate set"
for r in 1 2; do
VARIANTS=$V ROUNDS=9 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -7
done
