#!/bin/bash
mkdir -p gpurun_out
for c in 512; do
for k in nuc pc1 pc12 pc1x16; do
  if [ $k = none ]; then a=""; else a="--sort-chains $k"; fi
  timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline $a > gpurun_out/bench_sort_${k}.json 2> gpurun_out/bench_sort.err || tail -3 gpurun_out/bench_sort.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_sort_${k}.json").read().strip().splitlines()[-1])
print("chains $c sort $k value %.0f gfstack %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done; done
