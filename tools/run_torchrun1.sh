#!/bin/bash
# the driver's multi-rank launch line with one rank and the RCCL path forced on (init, barrier,
# all_reduce, all-gather of the stage transition)
mkdir -p gpurun_out
BEATAMD_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/torchrun1.json 2> gpurun_out/torchrun1.err
tail -c 700 gpurun_out/torchrun1.json; tail -3 gpurun_out/torchrun1.err
