#!/bin/bash
mkdir -p gpurun_out
BEATAMD_GS_DMA=4 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -3
for dma in 4 2; do
for c in 512 1024; do
  BEATAMD_GS_DMA=$dma timeout 200 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_ws.err | tail -1 > gpurun_out/bench_ws_${dma}_c${c}.json
  python -c "
import json; d=json.load(open('gpurun_out/bench_ws_${dma}_c${c}.json')); print('dma $dma chains $c value %.0f gfstack %.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))" || tail -3 gpurun_out/bench_ws.err
done; done
