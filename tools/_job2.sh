cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j36
for abl in 0 4 0 4; do
MS_ABL=$abl timeout 600 python tools/maxsim_ab.py --rounds 1 --steps 10 maxsim_wg=-1 2>&1 | grep "queries/s" | cut -c1-200 | sed "s/^/abl=$abl /"
done | tee gpurun_out/j36/abl.log
