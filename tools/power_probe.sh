#!/bin/bash
# what the chip draws and clocks at while the headline bench runs (is the screen kernel power-limited?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/power; mkdir -p $OUT
rocm-smi --showpower --showclocks --showmaxpower --showperflevel --showvoltage 2>&1 | head -60 > $OUT/idle.txt
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.02; done ) > $OUT/samples_bench.txt &
SMI=$!
python bench.py --steps ${STEPS:-600} --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} 2>/dev/null | tail -1 | cut -c1-300
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
cat $OUT/idle.txt | head -40
awk 'NR%8==0' $OUT/samples_bench.txt | head -60
