# one-step kernel timeline at an 8-GPU shard size (1.25M rows): every kernel of the last step with start offset + duration
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ROWS=${ROWS:-1250000}
rm -rf gpurun_out/tl; MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace -f csv -d gpurun_out/tl -- python bench.py --rows $ROWS --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/tl.log 2>&1
tail -1 gpurun_out/tl.log
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tl/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step = from the last k_prep_queries on
idx=[i for i,r in enumerate(rows) if 'k_prep_queries' in r['Kernel_Name']]
sel=rows[idx[-1]:]
t0=int(sel[0]['Start_Timestamp']); prev_end=t0
busy=0
for r in sel:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us  gap {(s-prev_end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
    prev_end=max(prev_end,e); busy+=e-s
print('span us', (prev_end-t0)/1e3, 'busy us', busy/1e3)
PY
