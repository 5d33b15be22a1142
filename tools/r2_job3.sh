#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j9
for g in 2 3; do for c in 1024 2048; do echo "== growth $g chunk0 $c =="; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --growth $g --chunk0 $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], 'big', r['launches'], r['avg_launch_ms'], r['kernel_ms_per_step'], 'all', r['all_screen_launches'], r['all_screen_kernels_ms_per_step'], d['extra']['candidates_per_query_per_step'], d['extra']['rescored_per_query_per_step'])"; done; done
echo "== sweep hit cost new kernel =="; VARIANTS=4436 SWEEP=1 timeout 300 tools/bin/screen_bench 9999872 1024 768 5 2>&1 | grep -v threshold
