cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcms; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -f csv -d $OUT -o sq -- python bench.py --workload maxsim --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -f csv -d $OUT -o m2 -- python bench.py --workload maxsim --steps 3 --warmup 1 --no-cpu-baseline > $OUT/m2.log 2>&1
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcms/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_maxsim16' in r['Kernel_Name']:
            agg['k_maxsim16'][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg['k_maxsim16'].items(): print(k, sum(v)/len(v), len(v))
PY
