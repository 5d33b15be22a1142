cd $GRAFT_REPO_ROOT
for rows in 1250000 2500000; do
for g in 2 3 4 5; do
for c0 in 512 1024; do
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --growth $g --chunk0 $c0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rows',$rows,'growth',$g,'chunk0',$c0,'ms',d['ms_per_step'],'qps',d['value'], 'retry', d['config'].get('retry_queries'), d['config'].get('fallback_queries'))"
done; done; done
