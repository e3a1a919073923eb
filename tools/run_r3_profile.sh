# round 3: rocprofv3 evidence for profiles/ -- the default bench command (kernel stats of every leg),
# the 512-chain nearest-neighbour run with its reuse-free streaming leg and the multilinear run
# (kernel stats + FETCH_SIZE / WRITE_SIZE in separate PMC passes), the geometry-mode stage
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variant-legs --no-narrow-leg --no-batch-leg"
prof() { tag=$1; shift; mkdir -p $O/$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/$tag/stats -o bench -- "$@" > $O/$tag/stats_run.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$tag/fetch -o bench -- "$@" > $O/$tag/fetch_run.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$tag/write -o bench -- "$@" > $O/$tag/write_run.log 2>&1
}
prof c512_nn $B
prof c512_ml $B --interp multilinear --no-streaming-leg
mkdir -p $O/default $O/geometry
timeout 900 rocprofv3 --kernel-trace --stats -d $O/default/stats -o bench -- python $R/bench.py --no-cpu-baseline > $O/default/stats_run.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/geometry/stats -o bench -- python $R/tools/geo_app.py 1024 200 > $O/geometry/stats_run.log 2>&1
S="python $R/tools/summarize_rocpd2.py"
$S $O/c512_nn $O/out r3_bench_c512_nn k_gfstack_ws "k_gfstack<0" k_fast_sweep k_accept > $O/sum_c512_nn.log 2>&1
$S $O/c512_ml $O/out r3_bench_c512_ml k_gfstack_cell k_gc_tables k_gc_order > $O/sum_c512_ml.log 2>&1
$S $O/default $O/out r3_bench_default k_quadform k_gfstack_cell > $O/sum_default.log 2>&1
$S $O/geometry $O/out r3_geometry_c1024 k_geom_los k_quadform_small k_draw_propose k_accept > $O/sum_geometry.log 2>&1
grep -h "^{\"metric" $O/*/stats_run.log > $O/out/r3_bench_lines_under_profiler.jsonl
tail -3 $O/default/stats_run.log $O/geometry/stats_run.log | cut -c1-300
cat $O/sum_*.log | grep -E "kernel\"|avg_us|corrected|write_bytes|no dispatch" | cut -c1-160
find $O -name "*.db" -size +2M -delete
find $O -name "*.csv" -size +2M -delete
du -sh $O
