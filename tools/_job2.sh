cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j24
timeout 900 python -m pytest tests/test_gpu_maxsim.py -m gpu -x -q > gpurun_out/j24/pytest.log 2>&1; tail -3 gpurun_out/j24/pytest.log
timeout 300 python tools/fuzz_parity.py --seconds 150 --only maxsim --seed 2424 > gpurun_out/j24/fuzz_maxsim.log 2>&1; tail -1 gpurun_out/j24/fuzz_maxsim.log | cut -c1-200
