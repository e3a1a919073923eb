mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -15
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --interp multilinear --chains 64 2>&1 | tail -1 > gpurun_out/bench_ml_c64.json
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --covariance toeplitz --chains 128 2>&1 | tail -1 > gpurun_out/bench_toep_c128.json
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --covariance toeplitz --chains 512 2>&1 | tail -1 > gpurun_out/bench_toep_c512.json
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains 256 2>&1 | tail -1 > gpurun_out/bench_c256.json
python - <<'PY'
import json,glob
for f in ['bench_ml_c64','bench_toep_c128','bench_toep_c512','bench_c256']:
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],1), round(d['roofline']['achieved'],1), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e: print(f, 'ERR', open('gpurun_out/%s.json'%f).read()[-600:])
PY
