mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prewhiten or pt_on or smc_on" 2>&1 | tail -5
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --covariance toeplitz --prewhiten 2>&1 | tail -1 > gpurun_out/bench_toep_pw.json
python - <<'PY'
import json
f='gpurun_out/bench_toep_pw.json'
try:
    d=json.load(open(f)); print(f, round(d['value'],1), d['setup_s'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
except Exception as e: print(f,'ERR',open(f).read()[-800:])
PY
