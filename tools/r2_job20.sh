#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tok in page text; do
docs=20000; [ $tok = text ] && docs=100000
rm -rf gpurun_out/ktm; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/ktm -o s -- python bench.py --workload maxsim --docs $docs --tokens $tok --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ktm.log 2>&1
echo "== $tok"; tail -1 gpurun_out/ktm.log | cut -c1-120
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ktm/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us avg', r['Percentage'])
PY
done
