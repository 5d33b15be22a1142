#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_maxsim.py tests/test_rerank.py tests/test_gpu_callers.py tests/test_gpu_sharded.py tests/test_multimodal.py -x -q 2>&1 | tail -2
timeout 400 python tools/fuzz_parity.py --seconds 150 --only maxsim --seed 4242 2>&1 | tail -1 | cut -c1-200
for shape in page text; do docs=100000; [ $shape = text ] && docs=1000000
python bench.py --workload maxsim --tokens $shape --docs $docs --steps 125 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$shape', 'ms/step', d['ms_per_step'], 'q/s', d['value'], 'screen ms', r['avg_launch_ms'], 'streamed GB/s', r['streamed_GBps'], 'alg frac', r['frac'], 'exact ms/step', r['exact_rescore_ms_per_step'], d['extra'])"
done
