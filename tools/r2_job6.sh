#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j12
echo "== sharded gpu test =="; timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/j12/pytest.log 2>&1; tail -15 gpurun_out/j12/pytest.log
for comm in torch lib; do echo "== bench --force-dist --comm $comm (1.25M rows = the 8-GPU shard) =="; timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --force-dist --comm $comm --rows 1250000 2>gpurun_out/j12/fd_$comm.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['parallelism'])"; tail -2 gpurun_out/j12/fd_$comm.err; done
echo "== plain 1.25M / 2.5M / 5M for the shard-size table =="; for n in 1250000 2500000 5000000 10000000; do for fd in "" "--force-dist"; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $fd --rows $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$fd', d['ms_per_step'], d['value'])"; done; done
