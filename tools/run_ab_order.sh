mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for o in 0 1; do timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --gf-order $o 2>&1 | tail -1 > gpurun_out/bench_order$o.json; done
for c in 32 512; do timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --gf-order 1 --chains $c 2>&1 | tail -1 > gpurun_out/bench_order1_c$c.json; done
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --gf-order 0 --chains 512 2>&1 | tail -1 > gpurun_out/bench_order0_c512.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_order*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value'],1), round(d['roofline']['achieved'],1), d['kernel_ms_per_step'])
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
PY
