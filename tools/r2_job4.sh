#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j10
echo "== new tests =="; timeout 900 python -m pytest tests/test_gpu_c2.py -x -q -m gpu -s > gpurun_out/j10/pytest_c2.log 2>&1; tail -15 gpurun_out/j10/pytest_c2.log
echo "== default bench =="; ( time timeout 600 python bench.py ) > gpurun_out/j10/bench_default.json 2>gpurun_out/j10/bench_default.err; tail -3 gpurun_out/j10/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/j10/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['hbm_view'], d.get('ndcg_at_10'), d['extra'].get('pcie_inclusive'), d.get('cpu_baseline'), d.get('cpu_baselines'))
PY
echo "== C2 bench: anisotropic, ip, k=100, 2M rows =="; timeout 600 python bench.py --data anisotropic --metric ip --k 100 --rows 2000000 --steps 10 --warmup 3 --cpu-sample-rows 250000 --cpu-sample-queries 512 > gpurun_out/j10/bench_c2.json 2>gpurun_out/j10/bench_c2.err; tail -3 gpurun_out/j10/bench_c2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/j10/bench_c2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['config']['workload'], d['extra'], d.get('ndcg_at_10'), d.get('cpu_baseline'))
PY
