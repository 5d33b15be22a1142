cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j28
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe || exit 1
for data in gauss absA offA abs off zero gauss; do
  ( for i in $(seq 1 120); do rocm-smi --showpower --showclocks 2>/dev/null | grep -a -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.03; done ) > gpurun_out/j28/smi_$data.txt &
  SMI=$!
  line=$(/tmp/mfma_power_probe i8 $data 4 | tail -1)
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  n=$(wc -l < gpurun_out/j28/smi_$data.txt)
  w=$(tail -n +$((n/3+1)) gpurun_out/j28/smi_$data.txt | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  m=$(tail -n +$((n/3+1)) gpurun_out/j28/smi_$data.txt | grep -a -o "([0-9]*Mhz)" | tr -d '()Mhz' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  echo "$line | socket $w W, $m MHz"
done 2>&1 | tee gpurun_out/j28/encoding_probe.txt
