cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j27
timeout 900 python tools/maxsim_ab.py --rounds 1 maxsim_wg_pipe=1 > gpurun_out/j27/ab.log 2>&1; tail -3 gpurun_out/j27/ab.log | cut -c1-400
