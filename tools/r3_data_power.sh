#!/bin/bash
# does the power-capped screen kernel's speed depend on the operand VALUES?  parked thresholds, N = 10 M, int8, library kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iautorag_research_amd/csrc -Itools/forms tools/screen_bench.hip -o /tmp/screen_bench || exit 1
for rep in 1 2; do for data in 1 2 3 4 5 6; do
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.05; done ) > gpurun_out/dp/smi_$data.txt &
SMI=$!
echo "DATA=$data: $(DATA=$data ROUNDS=60 VARIANTS=201000 /tmp/screen_bench 10000000 1024 768 | tail -1)"
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
echo "   smi: $(sort gpurun_out/dp/smi_$data.txt | uniq -c | sort -rn | head -3 | tr '\n' '|')"
done; done
echo "--- corpus Gaussian, query zeros / small"
for qd in 2 3; do echo "DATA=1 QDATA=$qd: $(DATA=1 QDATA=$qd ROUNDS=60 VARIANTS=201000 /tmp/screen_bench 10000000 1024 768 | tail -1)"; done
