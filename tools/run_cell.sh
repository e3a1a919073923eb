#!/bin/bash
# build check + correctness + timing of k_gfstack_cell in one GPU call: tools/gpu_cell.sh "<time_ml envs>"
set -e
cd /root/repo/beat_amd/csrc && make 2>&1 | grep -E "error|Error" && { echo BUILD FAILED; exit 1; } || true
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 600 -- "timeout 200 python tools/cell_check.py 2>&1 | grep -v amdgpu | tail -3; timeout 300 python tools/time_ml.py --envs \"$1\" 2>&1 | grep TIME" 2>&1 | tail -12
