#!/bin/bash
# first GPU session of the cell kernel: correctness script, then multilinear bench A/B on one box
mkdir -p gpurun_out
timeout 900 python tools/cell_check.py > gpurun_out/cell_check.log 2>&1
echo "cell_check rc=$?" >> gpurun_out/cell_check.log
if grep -q "CELL_CHECK OK" gpurun_out/cell_check.log; then
  B="--interp multilinear --steps 10 --warmup 3 --no-cpu-baseline --no-streaming-leg --no-batch-leg --no-narrow-leg"
  timeout 600 python bench.py $B > gpurun_out/bench_ml_cell.json 2> gpurun_out/bench_ml_cell.err
  BEATAMD_GC_SORT=0 timeout 600 python bench.py $B > gpurun_out/bench_ml_cell_nosort.json 2> gpurun_out/bench_ml_cell_nosort.err
  BEATAMD_GS_CELL=0 timeout 600 python bench.py $B > gpurun_out/bench_ml_dma.json 2> gpurun_out/bench_ml_dma.err
fi
tail -30 gpurun_out/cell_check.log
