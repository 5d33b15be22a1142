#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j8
echo "== gpu tests (search/kernels/fuzz) =="; timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/j8/pytest.log 2>&1; tail -5 gpurun_out/j8/pytest.log
for f in 0 1 0 1; do echo "== bench form $f =="; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --screen-form $f > gpurun_out/j8/bench_f$f.json 2>gpurun_out/j8/bench_f$f.err; cat gpurun_out/j8/bench_f$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_ms_per_step'], d['extra'])"; done
echo "== bf16 =="; for f in 0 1; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --screen bf16 --screen-form $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_ms_per_step'])"; done
