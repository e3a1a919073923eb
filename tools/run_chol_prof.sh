# per-kernel times of beatamd_chol_inverse_batch at 64 x 4096^2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cholprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o chol -- python $R/tools/time_chol.py 64 4096 > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-170
grep "hand-written" $O/run.log | tail -2
find $O -name "*kernel_trace.csv" -size +3M -delete
