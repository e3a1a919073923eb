#!/bin/bash
# full GPU suite + default bench + rocprofv3 evidence (kernel stats, FETCH/WRITE PMC passes)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/pytest_round.log
cat gpurun_out/pytest_round.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || tail -5 gpurun_out/bench_default.err
tail -c 2500 gpurun_out/bench_default.json
rm -rf gpurun_out/prof
sed -i 's/^BEATAMD_GF_KERNEL=0 BEATAMD_GF_ORDER=0.*$//' tools/run_profile.sh
bash tools/run_profile.sh > gpurun_out/profile.log 2>&1
tail -5 gpurun_out/profile.log
