cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j19
timeout 600 python -m pytest tests/test_gpu_maxsim.py -m gpu -x -q -k "packed or sixteen" > gpurun_out/j19/pytest.log 2>&1; tail -3 gpurun_out/j19/pytest.log
timeout 900 python tools/scratch/ms_ab.py > gpurun_out/j19/ms_ab.log 2>&1; tail -13 gpurun_out/j19/ms_ab.log
