# round 2: rocprofv3 evidence for profiles/ -- default bench (kernel stats + FETCH/WRITE PMC
# passes), multilinear, dense Toeplitz (quadform), and the small kernels
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-streaming-leg --no-narrow-leg --no-batch-leg"
prof() { tag=$1; shift; mkdir -p $O/$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/$tag/stats -o bench -- "$@" > $O/$tag/stats_run.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$tag/fetch -o bench -- "$@" > $O/$tag/fetch_run.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$tag/write -o bench -- "$@" > $O/$tag/write_run.log 2>&1
}
prof c512_nn $B
prof c512_ml $B --interp multilinear
prof c512_toeplitz $B --covariance toeplitz
prof c2048_nn $B --chains 2048 --steps 6
python $R/tools/summarize_rocpd2.py $O/c512_nn $O/out r2_bench_c512_nn k_gfstack k_gf_group_tables k_fast_sweep k_gf_tables > $O/sum_c512_nn.log 2>&1
python $R/tools/summarize_rocpd2.py $O/c512_ml $O/out r2_bench_c512_ml k_gfstack k_gf_group_tables > $O/sum_c512_ml.log 2>&1
python $R/tools/summarize_rocpd2.py $O/c512_toeplitz $O/out r2_bench_c512_toeplitz k_gfstack k_quadform > $O/sum_c512_toeplitz.log 2>&1
python $R/tools/summarize_rocpd2.py $O/c2048_nn $O/out r2_bench_c2048_nn k_gfstack > $O/sum_c2048.log 2>&1
cat $O/sum_*.log | tail -80
# keep the merged output small
find $O -name "*.db" -size +2M -delete
find $O -name "*.csv" -size +2M -delete
du -sh $O
