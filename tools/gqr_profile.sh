# rocprofv3 kernel stats of the GQR refinement kernels on the page-sized shapes of tests/test_gpu_gqr.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gqr; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/gqr -o gqr -- python -m pytest tests/test_gpu_gqr.py -q -k "bench_shaped or colbert_shape" > gpurun_out/gqr.log 2>&1
tail -2 gpurun_out/gqr.log
grep -i "k_gqr\|Name" gpurun_out/gqr/gqr_kernel_stats.csv | cut -c1-220
