cd $GRAFT_REPO_ROOT
for nt in 40 48; do BEATAMD_GS_NT=$nt timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shared or fused" 2>&1 | tail -1; done
for nt in 64 48 40 64 48 40; do BEATAMD_GS_NT=$nt timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains 256 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('nt=$nt c256', round(d['value'],1), round(d['kernel_ms_per_step']['gfstack'],3))"; done
for nt in 48 40; do BEATAMD_GS_NT=$nt timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains 512 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('nt=$nt c512', round(d['value'],1), round(d['kernel_ms_per_step']['gfstack'],3))"; done
