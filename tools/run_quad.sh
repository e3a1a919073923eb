mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
for c in 128 512; do timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --covariance toeplitz --chains $c 2>&1 | tail -1 > gpurun_out/bench_toep2_c$c.json; done
python - <<'PY'
import json
for c in (128,512):
    f='gpurun_out/bench_toep2_c%d.json'%c
    try:
        d=json.load(open(f)); q=d['kernel_ms_per_step']['quadform']; print(f, round(d['value'],1), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'TF=%.1f'%(c*1.0737/q))
    except Exception as e: print(f,'ERR',open(f).read()[-500:])
PY
