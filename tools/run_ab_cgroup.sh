mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -k "stack" 2>&1 | tail -3
for g in 32 64 128 256; do BEATAMD_GF_CGROUP=$g timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chains 512 2>&1 | tail -1 > gpurun_out/bench_cg${g}_c512.json; done
BEATAMD_GF_CGROUP=64 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --chains 128 2>&1 | tail -1 > gpurun_out/bench_cg64_c128.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_cg*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value'],1), round(d['roofline']['achieved'],1), round(d['kernel_ms_per_step']['gfstack'],3))
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
PY
