#!/bin/bash
# On the GPU box: tests, smoke, default bench line, rocprofv3 kernel stats and PMC passes -> gpurun_out/ev/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ev; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log | cut -c1-400
python bench.py --screen bf16 --no-cpu-baseline > $OUT/bench_bf16.log 2>&1; tail -1 $OUT/bench_bf16.log | cut -c1-200
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
ls $OUT/stats
bash tools/collect_traffic.sh > $OUT/traffic.log 2>&1; tail -5 $OUT/traffic.log
