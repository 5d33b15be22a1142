#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# third form, timing ablations: 202024 full | 202026 no fragment reads (MFMA on stale registers) + staging | 202040 reads + MFMA, no staging
# | 202042 MFMA only
for r in 1 2; do
VARIANTS=202024,202026,202040,202042 ROUNDS=11 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -4
done
