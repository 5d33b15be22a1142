#!/bin/bash
# timeline.sh <outfile> <bench args...>: every kernel of the bench's last step with start offset, gap and duration (rocprofv3 --kernel-trace)
out=$1; shift
T=$(mktemp -d /tmp/tl.XXXX)
MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace -f csv -d $T -- python bench.py "$@" --steps 3 --warmup 2 --no-extras --no-cpu-baseline > $T/log 2>&1
python - $T > $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_prep_queries' in r['Kernel_Name']]
sel = rows[idx[-1]:]
t0 = int(sel[0]['Start_Timestamp']); prev_end = t0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (s - t0) > 50e6: break
    print(f"{(s-t0)/1e3:9.1f} us  gap {(s-prev_end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
    prev_end = e
PY
rm -rf $T
