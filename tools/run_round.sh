#!/bin/bash
# full GPU suite + smoke + default bench (+ PROFILE=1: the round's rocprofv3 evidence, tools/run_r3_profile.sh)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/pytest_round.log
cat gpurun_out/pytest_round.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || tail -5 gpurun_out/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline'], d['cpu_baseline']['value'])"
if [ "$PROFILE" = "1" ]; then
  bash tools/run_r3_profile.sh > gpurun_out/profile.log 2>&1
  tail -3 gpurun_out/profile.log
fi
