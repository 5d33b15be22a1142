#!/bin/bash
# On the GPU box: bare int8 MFMA stream by operand-sharing pattern / instruction shape, with power and clock next to each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab; mkdir -p $OUT
P=tools/bin/mfma_power_probe
smi_poll() { ( for i in $(seq 1 $2); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.03; done ) > $1 & SMI=$!; }
smi_median() {
  local n=$(wc -l < $1)
  local w=$(tail -n +$((n/3+1)) $1 | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  local m=$(tail -n +$((n/3+1)) $1 | grep -a -o "([0-9]*Mhz)" | tr -d '()Mhz' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  echo "${w} W, ${m} MHz"
}
{
rocm-smi --showpower --showclocks 2>/dev/null | head -20
for f in i8 i8_nos i8_ab i8_16 i8; do
  smi_poll $OUT/smi_probe_$f.txt 40
  line=$(timeout 60 $P $f gauss 4 | tail -1)
  wait $SMI; echo "$line | $(smi_median $OUT/smi_probe_$f.txt)"
done
} 2>&1 | tee $OUT/probe_sharing.txt
