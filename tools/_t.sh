python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_config4.py -q -x -k "band or bidiag or split or n120 or mvn" 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 20 --repeats 1 --no-streaming-leg --no-batch-leg --no-narrow-leg --variant-legs toeplitz,default_config --full-json gpurun_out/bf_dc.json > gpurun_out/b_dc.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/bf_dc.json'))
for k in ("multilinear_dense_W","multilinear_banded_W"):
    x=d['default_config_leg'][k]; print(k, round(x['chain_steps_per_s']), round(x['ms_per_step'],3), {a:round(b,3) for a,b in (x.get('kernel_ms_per_step') or {}).items()})
t=d['toeplitz_leg']; print('toeplitz', round(t['chain_steps_per_s']), 'banded', round(t['banded']['chain_steps_per_s']))
PY
