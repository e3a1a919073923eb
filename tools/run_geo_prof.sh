# per-kernel times of the geometry-mode SMC stage (BASELINE configs[1]); outputs under gpurun_out/geoprof
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/geoprof; mkdir -p $R/gpurun_out/geoprof
cd /tmp && export TMPDIR=/tmp
timeout 120 python $R/tools/geo_app.py 1024 200 graph 2>&1 | grep -v amdgpu.ids
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/geoprof/stats -o geo -- python $R/tools/geo_app.py 1024 200 > $R/gpurun_out/geoprof/run.log 2>&1
cd $R/gpurun_out/geoprof
f=$(find . -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" | cut -c1-200
find . -name "*kernel_trace.csv" -size +4M -delete
grep "stage\|one batched" run.log
