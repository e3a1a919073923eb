#!/bin/bash
mkdir -p gpurun_out
for cg in 512 256; do
for c in 512 1024 2048; do
  BEATAMD_GS_CG=$cg timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_d5_${cg}_c${c}.json 2> gpurun_out/bench_dma.err || tail -3 gpurun_out/bench_dma.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_d5_${cg}_c${c}.json").read().strip().splitlines()[-1])
print("cg $cg chains $c value %.0f gfstack %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done; done
for c in 128 256; do
timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nn c$c', round(d['value']), d['roofline']['avg_launch_ms'])"
done
timeout 300 python bench.py --chains 256 --steps 4 --warmup 2 --no-cpu-baseline --interp multilinear | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ml c256', round(d['value']), d['roofline']['avg_launch_ms'])"
timeout 300 python bench.py --chains 512 --steps 4 --warmup 2 --no-cpu-baseline --interp multilinear | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ml c512', round(d['value']), d['roofline']['avg_launch_ms'])"
