#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --data anisotropic --metric ip --k 100 --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
python bench.py --k 100 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('k100', d['value'], d['ms_per_step'])"
