cd $GRAFT_REPO_ROOT
for k in 1 3 10 24; do
python bench.py --k $k --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('k',$k,'ms',d['ms_per_step'],'screen_ms',r.get('all_screen_kernels_ms_per_step'),'k256 ms',r.get('kernel_ms_per_step'),'launches/step',r.get('all_screen_launches')/20, {x:c[x] for x in c if 'cand' in x or 'resc' in x})"
done
