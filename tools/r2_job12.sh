#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# does the corpus stream's HBM latency bound the staging?  small corpora stay in the 256 MB Infinity Cache between launches
for n in 131072 262144 1048576 9999872; do
echo "== N=$n"; VARIANTS=4436,202024,202028 ROUNDS=21 timeout 300 tools/bin/screen_bench $n 1024 768 5 2>&1 | tail -3
done
