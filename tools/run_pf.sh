#!/bin/bash
# A/B of the L2-prefetch distance of k_gfstack_shared (BEATAMD_GS_PF patches ahead)
mkdir -p gpurun_out
for c in 512 256; do
for pf in 0 1 2 4; do
  BEATAMD_GS_PF=$pf timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_pf${pf}_c${c}.json 2> gpurun_out/bench_pf.err || tail -3 gpurun_out/bench_pf.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_pf${pf}_c${c}.json").read().strip().splitlines()[-1])
print("chains $c pf $pf value %.0f gfstack %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done; done
