cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/diagprof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/diagprof -- python3 $GRAFT_REPO_ROOT/scripts/chol_bench.py cfg2 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/diagprof -name "*kernel_stats.csv" | head -1)
grep "k_chol_diag\|k_gemm_nt" $f | cut -c1-120
