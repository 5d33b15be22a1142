#!/bin/bash
# clock and cycle count of the parked kernel (tools/screen_bench) for comparison with the in-search launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/j25; rm -rf $OUT; mkdir -p $OUT
VARIANTS=${V:-202024} ROUNDS=5 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d $OUT -o sb -- tools/bin/screen_bench 9999872 1024 768 5 > $OUT/sb.log 2>&1
tail -2 $OUT/sb.log
python - <<'PY'
import csv,glob,collections
kt={}
for r in csv.DictReader(open(glob.glob('gpurun_out/j25/**/*kernel_trace.csv',recursive=True)[0])):
    kt[r['Dispatch_Id']]=(r['Kernel_Name'][:40], int(r['End_Timestamp'])-int(r['Start_Timestamp']))
agg=collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob('gpurun_out/j25/**/*counter_collection.csv',recursive=True)[0])):
    agg[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
res=collections.defaultdict(list)
for d,c in list(agg.items()):
    n,dur=kt.get(d,('?',0))
    if 'screen256c' not in n: continue
    g=c.get('GRBM_GUI_ACTIVE',0)/8
    res[n].append((dur/1e3,g,g/dur if dur else 0,c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(c.get('SQ_BUSY_CU_CYCLES',1)*4),c.get('SQ_WAIT_ANY',0)/c.get('SQ_WAVE_CYCLES',1),c.get('SQ_WAIT_INST_ANY',0)/c.get('SQ_WAVE_CYCLES',1)))
for n,v in res.items():
    v=v[len(v)//2:]  # second half: clocks settled
    m=[sum(x[i] for x in v)/len(v) for i in range(6)]
    print(n, 'n',len(v),'dur_us %.0f Mcycles/xcd %.2f GHz %.3f mfma_busy %.3f wait_any %.3f wait_inst %.3f'%(m[0],m[1]/1e6,m[2],m[3],m[4],m[5]))
PY
