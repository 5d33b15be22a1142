cd $GRAFT_REPO_ROOT
for rows in 1250000 10000000; do for ra in 0 12 16 20 24 48; do for pf in 0 1; do
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --round-a $ra --prefilter16 $pf 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rows',$rows,'round_a',$ra,'pf',$pf,'ms',d['ms_per_step'],'qps',d['value'])"
done; done; done
