#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline --row-sharded-leg 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['extra'].get('row_sharded')))"
# the same under torch.distributed.run with one rank (the launcher path the driver uses)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline --row-sharded-leg --layout 1x1 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['extra'].get('row_sharded')))"
