#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/round_evidence.sh
timeout 400 python tools/fuzz_parity.py --seconds 300 2>&1 | tail -1 | cut -c1-300
