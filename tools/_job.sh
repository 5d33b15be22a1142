cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j6
timeout 900 python tools/scratch/ms_ab.py > gpurun_out/j6/ms_ab.log 2>&1; tail -12 gpurun_out/j6/ms_ab.log
