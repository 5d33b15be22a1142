#!/bin/bash
# small query blocks: k_screen_stream against k_screen (option screen_stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for B in 1 8 32 64 128; do
python bench.py --block $B --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B',$B,'ms',d['ms_per_step'],'qps',d['value'],'screen_ms',r.get('all_screen_kernels_ms_per_step'),'streamed',r.get('hbm_view',{}).get('streamed_GBps'))"
done
for B in 1 32 64; do
python bench.py --block $B --screen bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bf16 B',$B,'ms',d['ms_per_step'],'qps',d['value'],'screen_ms',r.get('all_screen_kernels_ms_per_step'),'streamed',r.get('hbm_view',{}).get('streamed_GBps'))"
done
timeout 400 python tools/fuzz_parity.py --seconds 300 --only single --seed 11 2>&1 | tail -1 | cut -c1-200
