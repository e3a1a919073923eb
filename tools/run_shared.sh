mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shared or stack" 2>&1 | tail -15
for k in 0 1; do BEATAMD_GF_KERNEL=$k timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains 128 2>&1 | tail -1 > gpurun_out/bench_k${k}_c128.json; done
for c in 256 512 1024; do BEATAMD_GF_KERNEL=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --chains $c 2>&1 | tail -1 > gpurun_out/bench_k1_c$c.json; done
BEATAMD_GF_KERNEL=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --chains 256 --interp multilinear 2>&1 | tail -1 > gpurun_out/bench_k1_ml_c256.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_k*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value'],1), round(d['roofline']['achieved'],1), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e: print(f, 'ERR', open(f).read()[-600:])
PY
