#!/bin/bash
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" 2> gpurun_out/var2.err | tail -1 > gpurun_out/var2_$tag.json; python -c "
import json; d=json.load(open('gpurun_out/var2_$tag.json')); print('$tag', round(d['value']), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})" || tail -3 gpurun_out/var2.err; }
run toep_c512 --chains 512 --covariance toeplitz
run toep_pw_c512 --chains 512 --covariance toeplitz --prewhiten
run ml_c512 --chains 512 --interp multilinear
run nn_c64 --chains 64
