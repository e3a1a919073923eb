#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -2
for cg in 512 256; do
for c in 512 1024 2048; do
  BEATAMD_GS_CG=$cg timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_d3_${cg}_c${c}.json 2> gpurun_out/bench_dma.err || tail -3 gpurun_out/bench_dma.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_d3_${cg}_c${c}.json").read().strip().splitlines()[-1])
print("cg $cg chains $c value %.0f gfstack %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done; done
