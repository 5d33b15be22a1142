cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j3
timeout 900 python -m pytest tests/test_gpu_maxsim.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/j3/pytest.log 2>&1; tail -5 gpurun_out/j3/pytest.log
timeout 900 python tools/scratch/ms_ab.py > gpurun_out/j3/ms_ab.log 2>&1; tail -12 gpurun_out/j3/ms_ab.log
