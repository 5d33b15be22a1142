#!/bin/bash
# MaxSim evidence (round 3): per-launch kernel timeline and rocprofv3 --stats of both store shapes at the bench's sizes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3ms; mkdir -p $OUT
for shape in text page; do
  docs=100000; [ $shape = page ] && docs=20000
  rm -rf $OUT/kt_$shape
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt_$shape -o ms -- python bench.py --workload maxsim --tokens $shape --docs $docs --steps 6 --warmup 2 --no-cpu-baseline > $OUT/line_$shape.log 2>&1
  tail -1 $OUT/line_$shape.log | cut -c1-600
  python - "$OUT/kt_$shape" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'mi355' in r['Kernel_Name'] or 'copy' in r['Kernel_Name'].lower() or 'fill' in r['Kernel_Name'].lower()]
# the last step = from the last k_maxsim16 launch on
idx=[i for i,r in enumerate(sel) if 'k_maxsim16' in r['Kernel_Name']]
sel=sel[idx[-1]:]
t0=int(sel[0]['Start_Timestamp']); prev=t0
for r in sel:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    print(f"{(st-t0)/1e3:9.1f} us gap {(st-prev)/1e3:6.1f} dur {(en-st)/1e3:8.1f}  {r['Kernel_Name'][:64]}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
    prev=max(prev,en)
PY
done
