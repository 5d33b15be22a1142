#!/bin/bash
# On the GPU box: interleaved A/B of k_screen256c against k_screen_rq (tools/screen_ab, built beforehand into tools/bin/)
# with socket power / clock next to a sustained run of each.  -> gpurun_out/ab/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab; mkdir -p $OUT
AB=tools/bin/screen_ab
ROWS=${ROWS:-4194304}
smi_poll() { ( for i in $(seq 1 $2); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.03; done ) > $1 & SMI=$!; }
smi_median() {
  local n=$(wc -l < $1)
  local w=$(tail -n +$((n/3+1)) $1 | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  local m=$(tail -n +$((n/3+1)) $1 | grep -a -o "([0-9]*Mhz)" | tr -d '()Mhz' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  echo "${w} W, ${m} MHz"
}
{
for data in 1 2; do
  echo "## DATA=$data (1 = Gaussian sigma 29 int8, 2 = zeros), rows $ROWS x 1024 queries x d 768"
  DATA=$data ROUNDS=${ROUNDS:-30} VARIANTS=${VARIANTS:-0,100,101,104,108,116,117} timeout 300 $AB $ROWS 1024 768
  for v in ${POWER_VARIANTS:-0 100}; do
    smi_poll $OUT/smi_${data}_${v}.txt 60
    DATA=$data VARIANTS=$v SECONDS_RUN=3 timeout 120 $AB $ROWS 1024 768 | tail -1
    wait $SMI; echo "   power / clock next to it: $(smi_median $OUT/smi_${data}_${v}.txt)"
  done
done
} 2>&1 | tee $OUT/ab_${TAG:-run}.txt
