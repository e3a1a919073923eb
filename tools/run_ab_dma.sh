#!/bin/bash
mkdir -p gpurun_out
BEATAMD_GS_DMA=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -4
for dma in 2 1; do
for c in 512 256; do
  BEATAMD_GS_DMA=$dma timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_d4_${dma}_c${c}.json 2> gpurun_out/bench_dma.err || tail -3 gpurun_out/bench_dma.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_d4_${dma}_c${c}.json").read().strip().splitlines()[-1])
print("dma $dma chains $c value %.0f gfstack %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done; done
