#!/bin/bash
# On the GPU box: PMC passes (separate, kernel-trace only) over the default bench command; summaries -> gpurun_out/traffic/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/traffic; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT -o fetch -- python bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT -o write -- python bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT -o tcc -- python bench.py $ARGS > $OUT/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT -o sq -- python bench.py $ARGS > $OUT/sq.log 2>&1
python tools/traffic_summary.py $OUT
