#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/scan; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/scan -o scan -- python tools/scan_bench.py ${ROWS:-2000000} 2>&1 | grep "scan path"
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/scan/scan_kernel_stats.csv")):
    if "k_scan" in r["Name"] or "k_prune" in r["Name"]:
        print(r["Name"][:60], "calls", r["Calls"], "avg us", float(r["AverageNs"])/1e3, "min", float(r["MinNs"])/1e3, "max", float(r["MaxNs"])/1e3)
PY
python -m pytest tests/test_gpu_search.py tests/test_gpu_c2.py tests/test_gpu_fuzz.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
