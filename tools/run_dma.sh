#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -5
for dma in 1 0; do
for c in 512 256 1024; do
  BEATAMD_GS_DMA=$dma timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dma${dma}_c${c}.json 2> gpurun_out/bench_dma.err || tail -3 gpurun_out/bench_dma.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_dma${dma}_c${c}.json").read().strip().splitlines()[-1])
print("dma $dma chains $c value %.0f gfstack %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done; done
BEATAMD_GS_DMA=1 timeout 300 python bench.py --chains 256 --steps 4 --warmup 2 --no-cpu-baseline --interp multilinear | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ml c256', round(d['value']), d['roofline']['avg_launch_ms'])"
