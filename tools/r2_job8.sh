#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j14
timeout 1500 python -m pytest tests/test_gpu_threads.py tests/test_gpu_search.py -x -q -m gpu > gpurun_out/j14/pytest.log 2>&1; tail -12 gpurun_out/j14/pytest.log
