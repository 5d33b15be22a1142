#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j11
echo "== gpu tests =="; timeout 1200 python -m pytest tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_c2.py -x -q -m gpu > gpurun_out/j11/pytest.log 2>&1; tail -5 gpurun_out/j11/pytest.log
echo "== fuzz 150 s (single) =="; timeout 400 python tools/fuzz_parity.py --seconds 150 --seed 21 --only single > gpurun_out/j11/fuzz.log 2>&1; tail -4 gpurun_out/j11/fuzz.log
echo "== scan path timing: 2M rows gaussian, path=scan, B=64 =="; python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0,'.')
import autorag_research_amd as pkg
from autorag_research_amd import synth
d=768; idx=pkg.Mi355Index(d)
for c in range(8):
    x=synth.gaussian_chunk(torch,c,250000,d,'cuda'); torch.cuda.synchronize(); idx.add_device(x.data_ptr(),250000); del x
idx.set_option("path","scan")
Q=np.random.default_rng(0).standard_normal((64,d)).astype(np.float32)
idx.search(Q[:8],10)
for B in (8,32,64):
    t=time.perf_counter(); idx.search(Q[:B],10); t=time.perf_counter()-t
    print(f"scan B={B}: {t*1e3:.2f} ms  -> {2e6*d*4*((B+31)//32)/t/1e12:.2f} TB/s of fp32 rows per pass-set")
PY
echo "== C2 bench: anisotropic, ip, k=100, 2M rows =="; timeout 600 python bench.py --data anisotropic --metric ip --k 100 --rows 2000000 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['extra'])"
