# round 2, GPU session A: all GPU tests, the default bench line, kernel variants, rocprof evidence
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
timeout 900 python tools/exp_variants.py $O/variants.jsonl > $O/variants.log 2>&1
tail -3 $O/variants.log
timeout 300 python tools/smc_app.py 512 30 > $O/smc_app.log 2>&1
tail -4 $O/smc_app.log
