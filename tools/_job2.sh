cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j26
timeout 600 python tools/maxsim_ab.py --docs-scale 0.1 --steps 5 --rounds 1 maxsim_wg_pipe=1 "maxsim_wg_pipe=0,maxsim_aligned=0" > gpurun_out/j26/ab.log 2>&1; tail -5 gpurun_out/j26/ab.log | cut -c1-330
