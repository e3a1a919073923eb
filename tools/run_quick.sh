#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -3
timeout 200 python bench.py --chains 512 --steps 6 --warmup 2 --no-cpu-baseline 2> gpurun_out/quick.err | tail -1 > gpurun_out/quick.json
python -c "
import json; d=json.load(open('gpurun_out/quick.json')); print('value %.0f' % d['value'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'acc', d['accept_rate_last_step'])" || tail -3 gpurun_out/quick.err
