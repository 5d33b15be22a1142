#!/bin/bash
# round-2 GPU job 1: second-form screen kernel -- micro-bench (old vs new, ablations), parity tests, A/B bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j1
echo "== screen_bench 4M =="; timeout 300 tools/bin/screen_bench 4194304 1024 768 5 > gpurun_out/j1/sb_4m.log 2>&1; tail -40 gpurun_out/j1/sb_4m.log
echo "== screen_bench 10M i8 sweep =="; VARIANTS=1256,1300,1256,1300 SWEEP=1 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 > gpurun_out/j1/sb_10m.log 2>&1; tail -40 gpurun_out/j1/sb_10m.log
echo "== gpu tests (search/kernels/fuzz) =="; timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/j1/pytest.log 2>&1; tail -5 gpurun_out/j1/pytest.log
for f in 0 1 0 1; do echo "== bench form $f =="; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --screen-form $f > gpurun_out/j1/bench_f$f.json 2>gpurun_out/j1/bench_f$f.err; cat gpurun_out/j1/bench_f$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_ms_per_step'], d['extra'])"; done
