#!/bin/bash
# On the GPU box: a round's evidence, by section -> gpurun_out/ev/ (copy what is to be judged into profiles/rNN_*).
#   bash tools/round_evidence.sh [section ...]      (no argument = the default set)
# Sections (each writes gpurun_out/ev/<section>*):
#   tests      pytest -m gpu + smoke()
#   bench      the default bench line (what the driver runs) + the bf16-screen line
#   stats      rocprofv3 --kernel-trace --stats of the bench command
#   traffic    PMC passes (separate, kernel-trace only) of the bench command: FETCH_SIZE, WRITE_SIZE, L2 hit, SQ busy / LDS
#   timelines  one-step kernel timelines at N = 10 M and at the 8-way shard size
#   k100       BASELINE config 2's limit (top-100) at the headline corpus: bench lines at N = 10 M / 1.25 M, one-pass timelines,
#              the round-5 schedule (prune_wide 0) and the unsynchronised queue flush (screen_flush_sync 0) beside them
#   shards     pass time at the 1/2/4/8-GPU shard sizes, plain and through the all-gather + merge path (world 1)
#   c2         config C2 stand-in (anisotropic, inner product, k = 100): line + kernel stats
#   maxsim     the two MaxSim stores as bench lines of their own + kernel stats + PMC (FETCH_SIZE; SQ busy)
#   power      socket power / clock next to a 600-step bench run
#   mx         bare MFMA stream by operand format (int8, bf16, MX fp8 / fp6 / fp4) with power, + the int8 gather second stage
#   kstep      k_screen256c vs k_screen_rq interleaved + k_screen_rq timing builds (no LDS-DMA / fragment reads / tests / barrier), Gaussian operands and zeros, with power
#   barrier    grid-barrier cost in the screen kernel's geometry vs a dependent launch
#   fuzz       tools/fuzz_parity.py campaign (FUZZ_SECONDS, default 600)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ev; mkdir -p $OUT
SECTIONS="${@:-tests bench stats traffic timelines k100 shards c2 maxsim}"
BENCH_QUICK="--no-cpu-baseline --no-extras"

smi_poll() {  # smi_poll <file> <samples>: "(<MHz>Mhz) <W>" per line
  ( for i in $(seq 1 $2); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.03; done ) > $1 &
  SMI=$!
}
smi_median() {  # smi_median <file>: "<W> W, <MHz> MHz" over the samples behind the first third (the governor's ramp)
  local n=$(wc -l < $1)
  local w=$(tail -n +$((n/3+1)) $1 | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  local m=$(tail -n +$((n/3+1)) $1 | grep -a -o "([0-9]*Mhz)" | tr -d '()Mhz' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  echo "${w} W, ${m} MHz"
}
timeline() {  # timeline <rows> <outfile>: every kernel of the bench's last step with start offset, gap and duration
  rm -rf $OUT/tl; MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace -f csv -d $OUT/tl -- python bench.py --rows $1 --steps 3 --warmup 2 $BENCH_QUICK > $OUT/tl.log 2>&1
  python - $OUT/tl > $2 <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_prep_queries' in r['Kernel_Name']]
sel = rows[idx[-1]:]
t0 = int(sel[0]['Start_Timestamp']); prev_end = t0; busy = 0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (s - t0) > 50e6: break
    print(f"{(s-t0)/1e3:9.1f} us  gap {(s-prev_end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
    prev_end = max(prev_end, e); busy += e - s
print('span us', (prev_end - t0) / 1e3, 'busy us', busy / 1e3)
PY
  tail -3 $2
}

for sec in $SECTIONS; do case $sec in
tests)
  timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
bench)
  python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default_line.json; cut -c1-400 $OUT/bench_default_line.json
  python bench.py --screen bf16 $BENCH_QUICK --traffic > $OUT/bench_bf16.log 2>&1; tail -1 $OUT/bench_bf16.log > $OUT/bench_bf16_line.json; cut -c1-200 $OUT/bench_bf16_line.json ;;
stats)
  # the default command's steps / warm-up (10 / 2), without its untimed extras (they launch the same kernel on other workloads);
  # the profiled run's own line is kept next to the summary: its roofline.avg_launch_ms is the HIP-event figure of the very launches
  # the kernel_stats row averages (+ the 2 warm-up steps, whose first launch runs at cold clocks)
  rm -rf $OUT/stats; MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python bench.py --steps 10 --warmup 2 $BENCH_QUICK > $OUT/stats.log 2>&1
  grep '^{"metric"' $OUT/stats.log | tail -1 > $OUT/stats_line.json   # (rocprofv3 prints behind the bench line)
  find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -4
  python - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
line = json.loads(open(out + "/stats_line.json").read())
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
row = [r for r in csv.DictReader(open(f)) if "k_screen_rq" in r["Name"] or "k_screen256c" in r["Name"]][0]
print("rocprofv3 average", round(float(row["AverageNs"]) / 1e6, 4), "ms over", row["Calls"], "launches | the same run's HIP events:",
      line["roofline"]["avg_launch_ms"], "ms over", line["roofline"]["launches"], "timed launches")
PY
  ;;
traffic)
  T=$OUT/traffic; rm -rf $T; mkdir -p $T; ARGS="--steps 3 --warmup 1 $BENCH_QUICK"
  MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $T -o fetch -- python bench.py $ARGS > $T/fetch.log 2>&1
  MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $T -o write -- python bench.py $ARGS > $T/write.log 2>&1
  MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $T -o tcc -- python bench.py $ARGS > $T/tcc.log 2>&1
  MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $T -o sq -- python bench.py $ARGS > $T/sq.log 2>&1
  python tools/traffic_summary.py $T | tail -12 ;;
timelines)
  timeline 10000000 $OUT/timeline_10m.txt
  timeline 1250000 $OUT/timeline_1250k.txt ;;
k100)
  for rows in 10000000 1250000; do
    for o in "" "--opt screen_flush_sync=0" "--opt prune_wide=0"; do
      python bench.py --k 100 --rows $rows --steps 20 --warmup 3 $BENCH_QUICK $o 2>/dev/null | tail -1 | python -c "
import sys, json; r = json.loads(sys.stdin.read()); e = r['extra']
print('k100 rows $rows opts [$o] ms_per_step', r['ms_per_step'], 'screens', r['roofline']['all_screen_kernels_ms_per_step'], 'launches/pass', r['roofline']['all_screen_launches'] / r['steps'], 'cand', e['candidates_per_query_per_step'], 'resc', e['rescored_per_query_per_step'], 'retry', e['retry_queries'], 'fallback', e['fallback_queries'])"
    done
  done | tee $OUT/k100_lines.txt
  bash tools/timeline.sh $OUT/timeline_k100.txt --k 100
  bash tools/timeline.sh $OUT/timeline_k100_1250k.txt --k 100 --rows 1250000
  bash tools/timeline.sh $OUT/timeline_k100_r5_schedule.txt --k 100 --opt prune_wide=0
  head -20 $OUT/timeline_k100.txt ;;
shards)
  export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
  for rows in 10000000 5000000 2500000 1250000; do
    a=$(python bench.py --rows $rows --steps 30 --warmup 3 $BENCH_QUICK 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    b=$(python bench.py --rows $rows --steps 30 --warmup 3 $BENCH_QUICK --force-dist 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rows $rows  ms_per_pass $a  with_gather_and_merge $b"
  done | tee $OUT/shard_sizes.txt
  unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK ;;
c2)
  C2="--data anisotropic --metric ip --k 100 --rows 2681468"   # (BEIR nq's row count: BASELINE config 2)
  python bench.py $C2 --steps 20 --warmup 3 $BENCH_QUICK --traffic > $OUT/bench_c2.log 2>&1; tail -1 $OUT/bench_c2.log > $OUT/bench_c2_line.json; cut -c1-300 $OUT/bench_c2_line.json
  rm -rf $OUT/stats_c2; MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_c2 -o stats -- python bench.py $C2 --steps 5 --warmup 2 $BENCH_QUICK > $OUT/stats_c2.log 2>&1 ;;
maxsim)
  for shape in text page; do
    docs=1000000; [ $shape = page ] && docs=100000
    ARGS="--workload maxsim --tokens $shape --docs $docs --warmup 2 --no-cpu-baseline"
    python bench.py $ARGS --steps 63 > $OUT/maxsim_line_$shape.json 2> $OUT/maxsim_line_$shape.err; cut -c1-600 $OUT/maxsim_line_$shape.json
    rm -rf $OUT/maxsim_st_$shape $OUT/maxsim_pmc_$shape
    MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --stats -f csv -d $OUT/maxsim_st_$shape -o ms -- python bench.py $ARGS --steps 12 --no-extras > $OUT/maxsim_st_$shape.log 2>&1
    MI355DR_BENCH_PMC=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/maxsim_pmc_$shape -o sq -- python bench.py $ARGS --steps 6 --no-extras > $OUT/maxsim_pmc_$shape.log 2>&1
    python tools/pmc_summary.py $OUT/maxsim_pmc_$shape k_maxsim16 > $OUT/maxsim_pmc_${shape}_summary.json 2>/dev/null; cut -c1-600 $OUT/maxsim_pmc_${shape}_summary.json
  done ;;
power)
  smi_poll $OUT/power_samples.txt 500
  python bench.py --steps 600 --warmup 3 $BENCH_QUICK 2>/dev/null | tail -1 | cut -c1-300 | tee $OUT/power_bench_line.txt
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  echo "600-step bench: $(smi_median $OUT/power_samples.txt)" | tee $OUT/power_headline.txt ;;
mx)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iautorag_research_amd/csrc tools/mfma_power_probe.hip -o /tmp/mfma_power_probe || exit 1
  { echo "# bare MFMA stream: 256 workgroups x 8 waves (2 per SIMD), operands in registers, 4 accumulators; ${SECS:-4} s per run;"
    echo "# rocm-smi polled next to each run, medians behind the first third of the samples"
    rocm-smi --showmaxpower 2>/dev/null | grep -E "Max Graphics" | head -1
    for data in gauss zero; do for fmt in i8 bf16 fp8 fp6 fp4; do
      smi_poll $OUT/smi_mx.txt 200
      line=$(/tmp/mfma_power_probe $fmt $data ${SECS:-4} | tail -1)
      kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
      echo "$line | socket $(smi_median $OUT/smi_mx.txt)"
    done; done
    echo; echo "# int8 second stage by GATHER: random rows of a 10 M x 768 B int8 shadow, v_dot4_i32_i8, 16 lanes per row"
    for c in 7000 700 100; do /tmp/mfma_power_probe gather 10000000 1024 $c | tail -1; done; } 2>&1 | tee $OUT/mx_probe.txt ;;
kstep)
  # the A/B of the two large-block int8 screens + the timing builds of the new one, each with power / clock next to a sustained run
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iautorag_research_amd/csrc -Itools tools/screen_ab.hip -o /tmp/screen_ab || exit 1
  { echo "# k_screen256c (variant 0: 256 x 256 tile, both operands through the LDS) against k_screen_rq (variant 100: 128 rows x 256"
    echo "# queries, query operand resident in registers), int8, ${KROWS:-10000000} rows x 1024 queries x d 768, thresholds parked; timing builds of"
    echo "# k_screen_rq: 101 no fragment reads | 104 no tests | 108 no barrier | 116 no LDS-DMA | 117 no LDS-DMA, no fragment reads"
    echo "# (results of the ablated builds are garbage; without the LDS-DMA the ring is never filled: those builds multiply zeros)"
    echo "# 132 no vmcnt at the hand-over | 8292 block tests in one piece | 4196 no drift limiter"
    echo "# Operands written by a grid-stride fill and CHECKED (tools/screen_ab: a refused one-thread-per-byte fill left the runs above"
    echo "# 5.59 M rows of the earlier tables partly unwritten -- their absolute times were too fast; profiles/r05_harness_fill_fix.txt)"
    for data in 3 2; do
      echo "=== DATA=$data ($([ $data = 3 ] && echo 'true Gaussian int8, rows sigma 31.4, queries sigma 40: what the shadows hold' || echo zeros)) ==="
      echo "--- interleaved (every variant once per round, 24 rounds)"
      DATA=$data ROUNDS=24 VARIANTS=0,100,101,104,108,116,117,132,8292,4196 /tmp/screen_ab ${KROWS:-10000000} 1024 768 | grep -E "^variant|candidate set"
      for v in 0 100; do
        smi_poll $OUT/smi_kstep.txt 14
        line=$(DATA=$data VARIANTS=$v SECONDS_RUN=4 /tmp/screen_ab ${KROWS:-10000000} 1024 768 | tail -1)
        wait $SMI 2>/dev/null
        echo "sustained    $line | $(smi_median $OUT/smi_kstep.txt)"
      done
    done; } 2>&1 | tee $OUT/kstep_ab.txt ;;
barrier)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grid_barrier_probe.hip -o /tmp/grid_barrier_probe && timeout 120 /tmp/grid_barrier_probe | tee $OUT/grid_barrier.txt ;;
fuzz)
  timeout $(( ${FUZZ_SECONDS:-600} + 120 )) python tools/fuzz_parity.py --seconds ${FUZZ_SECONDS:-600} --seed ${FUZZ_SEED:-4} > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log | cut -c1-300 ;;
*) echo "unknown section $sec" ;;
esac; done
