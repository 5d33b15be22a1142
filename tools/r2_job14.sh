#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "=== OLD"; timeout 120 ab_old/tools/bin/screen_trace 4194304 2>&1 | tail -60
echo "=== NEW"; timeout 120 tools/bin/screen_trace 4194304 2>&1 | tail -60
