cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace -f csv -d gpurun_out/kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/kt.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/kt/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'k_screen' in r['Kernel_Name'] or 'k_prune' in r['Kernel_Name']]
for r in sel[-40:]:
    print(r['Kernel_Name'][:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us', r.get('Grid_Size_X', r.get('Grid_Size','')))
PY
