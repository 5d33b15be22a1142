# A/B of the bf16 second screen inside k_prune (option "prefilter16") at the full and the 8-GPU shard size
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
for rows in 1250000 10000000; do for pf in 0 1; do
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --prefilter16 $pf 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('rows',$rows,'prefilter16',$pf,'ms',d['ms_per_step'],'qps',d['value'], {k:c[k] for k in c if 'rescor' in k or 'candid' in k or 'retry' in k or 'fallback' in k})"
done; done
