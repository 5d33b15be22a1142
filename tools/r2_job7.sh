#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j13
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/j13/pytest_gpu.log 2>&1; tail -12 gpurun_out/j13/pytest_gpu.log
