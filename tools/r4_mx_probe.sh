#!/bin/bash
# (a) W / GHz / TOP/s of the bare matrix pipe by operand format (int8, bf16, MX fp8 / fp6 / fp4), Gaussian
# operands and zeros, rocm-smi polled next to each run; (c) the int8 gather-based second stage a loose pre-screen would need.
# (b) -- what such a pre-screen would have to admit -- is CPU work: tools/mx_bound_probe.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/mx; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe || exit 1
SECS=${SECS:-4}
{
echo "# bare MFMA stream, 256 workgroups x 8 waves (2 per SIMD), operands in registers, 4 accumulators; ${SECS} s per run"
echo "# rocm-smi polled every ~60 ms next to the run; medians over the samples after the first second"
rocm-smi --showmaxpower 2>/dev/null | grep -E "Max Graphics" | head -1
for data in gauss zero; do for fmt in i8 bf16 fp8 fp6 fp4; do
  ( for i in $(seq 1 200); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.03; done ) > $OUT/smi_${fmt}_${data}.txt &
  SMI=$!
  line=$(/tmp/mfma_power_probe $fmt $data $SECS | tail -1)
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  # samples: "<W> <level> (<MHz>Mhz)"; drop the first third (governor ramp)
  n=$(wc -l < $OUT/smi_${fmt}_${data}.txt)
  w=$(tail -n +$((n/3+1)) $OUT/smi_${fmt}_${data}.txt | awk '{print $1}' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  mhz=$(tail -n +$((n/3+1)) $OUT/smi_${fmt}_${data}.txt | grep -o "([0-9]*Mhz)" | tr -d '()Mhz' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  echo "$line | socket ${w} W, sclk ${mhz} MHz (median of $((n - n/3)) samples)"
done; done
echo
echo "# (c) int8 second stage by GATHER: random rows of a 10 M x 768 B int8 shadow, v_dot4_i32_i8, 16 lanes per row"
/tmp/mfma_power_probe gather 10000000 1024 7000
/tmp/mfma_power_probe gather 10000000 1024 700
/tmp/mfma_power_probe gather 10000000 1024 100
} 2>&1 | tee $OUT/r04_mx_probe.txt
