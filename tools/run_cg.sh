#!/bin/bash
mkdir -p gpurun_out
for cg in 64 128 256; do
  BEATAMD_GS_CG=$cg timeout 300 python bench.py --chains 512 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_gscg${cg}.json 2> gpurun_out/bench_cg.err || tail -3 gpurun_out/bench_cg.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_gscg${cg}.json").read().strip().splitlines()[-1])
print("gs_cg $cg value %.0f gfstack %.3f ms grouptables %.3f" % (d["value"], d["roofline"]["avg_launch_ms"], d["kernel_ms_per_step"].get("grouptables",0)))
PY
done
