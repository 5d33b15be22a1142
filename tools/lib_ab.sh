# A/B of two builds of libmi355dr.so in one GPU session (box-to-box variance is +-3 %): libmi355dr_old.so vs the current one
cd $GRAFT_REPO_ROOT/autorag_research_amd
cp libmi355dr.so /tmp/new.so; cp libmi355dr_old.so /tmp/old.so
cd ..
for rep in 1 2 3; do for v in old new; do
cp /tmp/$v.so autorag_research_amd/libmi355dr.so
for rows in 10000000 1250000; do
python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v rows',$rows,'ms',d['ms_per_step'],'screen_ms',d['roofline'].get('all_screen_kernels_ms_per_step'))"
done; done; done
cp /tmp/new.so autorag_research_amd/libmi355dr.so
python -m pytest tests/test_gpu_search.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
