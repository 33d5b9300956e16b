cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r03d
rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > "$out/bench_under_profiler.json" 2> "$out/stats.err"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-150
