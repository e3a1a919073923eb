#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -3
for c in 64 100 130 192 300 512 600 700 1024; do
  timeout 200 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline 2> gpurun_out/small.err | tail -1 > gpurun_out/small_c${c}.json
  python -c "
import json; d=json.load(open('gpurun_out/small_c${c}.json')); print('chains $c value %.0f gfstack %.3f ms step %.3f' % (d['value'], d['roofline']['avg_launch_ms'], d['ms_per_step']))" || tail -3 gpurun_out/small.err
done
