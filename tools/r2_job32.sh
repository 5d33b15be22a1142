#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1
bash tools/r2_job31.sh
timeout 200 python tools/fuzz_parity.py --seconds 100 --only single 2>&1 | tail -1 | cut -c1-160
