#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j13; mkdir -p $OUT; rm -f $OUT/*.json
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for r in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/a_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j13/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms_per_step'], r['all_screen_kernels_ms_per_step'])
PY
timeout 400 python tools/fuzz_parity.py --seconds 300 2>&1 | tail -1 | cut -c1-200
