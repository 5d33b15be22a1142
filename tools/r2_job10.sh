#!/bin/bash
# per-group int8 steps: correctness first, then the bench line, then hit-cost sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j10; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py -m gpu -x -q > $OUT/t1.log 2>&1; tail -15 $OUT/t1.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_kernels.py --deselect tests/test_gpu_search.py > $OUT/t2.log 2>&1; tail -8 $OUT/t2.log
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $OUT/b1.log 2>&1; tail -1 $OUT/b1.log | cut -c1-1800
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --force-dist > $OUT/b2.log 2>&1; tail -1 $OUT/b2.log | cut -c1-300
timeout 300 python tools/fuzz_parity.py --seconds 120 > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
