set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-r2h}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
timeout 600 python bench.py --steps 10 --warmup 2 --covariance toeplitz --no-cpu-baseline > $O/bench_toeplitz.json 2> $O/bench_toeplitz.err
timeout 600 python bench.py --steps 10 --warmup 2 --interp multilinear --no-cpu-baseline > $O/bench_ml.json 2> $O/bench_ml.err
bash tools/run_r2_profile.sh > $O/profile.log 2>&1
tail -30 $O/profile.log
timeout 300 python tools/smc_app.py 512 30 > $O/smc_app.log 2>&1
tail -4 $O/smc_app.log
