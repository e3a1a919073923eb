#!/bin/bash
# ablations of k_gfstack_cell (GC_ABLATIONS build): timing only
mkdir -p gpurun_out
B="--interp multilinear --steps 6 --warmup 2 --no-cpu-baseline --no-streaming-leg --no-batch-leg --no-narrow-leg"
for v in 0 1 2 3 4; do
  BEATAMD_GC_VAR=$v timeout 300 python bench.py $B > gpurun_out/abl_$v.json 2> gpurun_out/abl_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/abl_$v.json'));print('var',$v,round(d['kernel_ms_per_step']['gfstack'],2))"
done
