# per-kernel times of the pre-whitening of the 62.9 GB library (bench prewhitened leg)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/whitenprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/s -o w -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-streaming-leg --no-batch-leg --no-narrow-leg --variant-legs prewhitened > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-150
f=$(find $O -name "*memory_copy_stats.csv" | head -1)
[ -n "$f" ] && head -6 "$f" | cut -c1-150
grep -o '"whitening_s": [0-9.]*' $O/run.log
find $O -name "*trace.csv" -size +3M -delete
