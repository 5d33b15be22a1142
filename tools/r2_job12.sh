#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 202024 third form | 202056 timing only: wait for the pieces of the K-step BEFORE (vmcnt 9) | 202120 timing only: no vmcnt wait at all
for r in 1 2; do
VARIANTS=4436,202024,202056,202120 ROUNDS=11 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | tail -4
done
