cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j29
timeout 1300 python tools/fuzz_parity.py --seconds 1200 --seed 97531 > gpurun_out/j29/fuzz.log 2>&1; tail -1 gpurun_out/j29/fuzz.log | cut -c1-300
