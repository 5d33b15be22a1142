cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j4
timeout 900 python -m pytest tests/test_gpu_maxsim.py tests/test_gpu_sharded.py tests/test_rerank.py tests/test_gpu_callers.py tests/test_gpu_gqr.py -m gpu -x -q > gpurun_out/j4/pytest.log 2>&1; tail -5 gpurun_out/j4/pytest.log
timeout 400 python tools/fuzz_parity.py --seconds 240 --only maxsim --seed 404 > gpurun_out/j4/fuzz_maxsim.log 2>&1; tail -4 gpurun_out/j4/fuzz_maxsim.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grid_barrier_probe.hip -o /tmp/gbp && timeout 120 /tmp/gbp > gpurun_out/j4/grid_barrier.txt 2>&1; cat gpurun_out/j4/grid_barrier.txt
timeout 900 python tools/scratch/ms_ab.py > gpurun_out/j4/ms_ab.log 2>&1; tail -12 gpurun_out/j4/ms_ab.log
