#!/bin/bash
mkdir -p gpurun_out
for pf in 2 0; do
BEATAMD_GS_PF=$pf timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -2
done
for pf in 0 1 2 3 4; do
for c in 512; do
  BEATAMD_GS_PF=$pf timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dpf${pf}_c${c}.json 2> gpurun_out/bench_dma.err || tail -3 gpurun_out/bench_dma.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_dpf${pf}_c${c}.json").read().strip().splitlines()[-1])
print("pf $pf chains $c value %.0f gfstack %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done; done
