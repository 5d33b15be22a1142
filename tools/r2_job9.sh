#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j15
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_c2.py tests/test_gpu_maxsim.py -x -q -m gpu > gpurun_out/j15/pytest.log 2>&1; tail -6 gpurun_out/j15/pytest.log
timeout 300 python tools/fuzz_parity.py --seconds 90 --seed 33 --only single > gpurun_out/j15/fuzz.log 2>&1; tail -2 gpurun_out/j15/fuzz.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'], d['extra'])"; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --rows 1250000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('1.25M', d['ms_per_step'], d['value'], r['kernel_ms_per_step'])"
