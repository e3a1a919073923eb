cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k config4 2>&1 | tail -5
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --chains 4096 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('c4096', round(d['value'],1), d['kernel_ms_per_step'], d['accept_rate_last_step'])"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('default', round(d['value'],1), d['kernel_ms_per_step'], d['accept_rate_last_step'], d['roofline'])"
