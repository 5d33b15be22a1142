# queries-per-pass sweep at N = 10 M (the table of DESIGN 4.1)
cd $GRAFT_REPO_ROOT
for B in 1 32 128 256 512 1024; do
python bench.py --block $B --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B',$B,'ms',d['ms_per_step'],'qps',d['value'],'screen_ms',r.get('all_screen_kernels_ms_per_step'),'tops',r.get('achieved'),'hbm',r.get('hbm_view'))"
done
