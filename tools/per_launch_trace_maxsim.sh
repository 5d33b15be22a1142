cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ktm; rocprofv3 --kernel-trace -f csv -d gpurun_out/ktm -- python bench.py --workload maxsim --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ktm.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ktm/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'mi355' in r['Kernel_Name'] or 'copy' in r['Kernel_Name'].lower() or 'fill' in r['Kernel_Name'].lower()]
t0=None
for r in sel[-34:]:
    st=int(r['Start_Timestamp']); 
    if t0 is None: t0=st
    print(f"{(st-t0)/1e3:9.1f}", r['Kernel_Name'][:50], (int(r['End_Timestamp'])-st)/1e3, 'us', r.get('Grid_Size_X', r.get('Grid_Size','')))
PY
