#!/bin/bash
# On the GPU box: the round's evidence in one job -> gpurun_out/ev/ (copy what is to be judged into profiles/rNN_*).
#   tests + smoke, default bench line, bf16 line, rocprofv3 kernel stats of the bench, PMC traffic passes, one-step kernel
#   timelines at 10 M rows and at the 8-way shard size, shard-size table through the gather + merge path, config C2.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ev; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default_line.json; cut -c1-500 $OUT/bench_default_line.json
python bench.py --screen bf16 --no-cpu-baseline --no-extras > $OUT/bench_bf16.log 2>&1; tail -1 $OUT/bench_bf16.log > $OUT/bench_bf16_line.json; cut -c1-200 $OUT/bench_bf16_line.json
rm -rf $OUT/stats; rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
ls $OUT/stats
bash tools/collect_traffic.sh > $OUT/traffic.log 2>&1; tail -5 $OUT/traffic.log
ROWS=10000000 BENCH_ARGS=--no-extras bash tools/step_timeline.sh > $OUT/timeline_10m.txt 2>&1; tail -3 $OUT/timeline_10m.txt
ROWS=1250000 BENCH_ARGS=--no-extras bash tools/step_timeline.sh > $OUT/timeline_1250k.txt 2>&1; tail -3 $OUT/timeline_1250k.txt
bash tools/shard_sizes.sh > $OUT/shard_sizes.txt 2>&1; cat $OUT/shard_sizes.txt
# config C2 (BEIR nq / bge-base stand-in): line + kernel stats
python bench.py --data anisotropic --metric ip --k 100 --rows 2000000 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_c2.log 2>&1; tail -1 $OUT/bench_c2.log > $OUT/bench_c2_line.json; cut -c1-300 $OUT/bench_c2_line.json
rm -rf $OUT/stats_c2; rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_c2 -o stats -- python bench.py --data anisotropic --metric ip --k 100 --rows 2000000 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats_c2.log 2>&1
