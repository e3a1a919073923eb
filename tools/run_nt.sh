mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
BEATAMD_GS_NT=32 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shared or fused" 2>&1 | tail -4
for nt in 64 32 64 32; do BEATAMD_GS_NT=$nt timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains 256 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('nt=$nt', round(d['value'],1), round(d['kernel_ms_per_step']['gfstack'],3))"; done
for c in 128 512; do BEATAMD_GS_NT=32 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains $c 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('nt=32 c$c', round(d['value'],1), round(d['kernel_ms_per_step']['gfstack'],3))"; done
BEATAMD_GS_NT=32 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains 256 --interp multilinear 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('nt=32 ml c256', round(d['value'],1), round(d['kernel_ms_per_step']['gfstack'],3))"
