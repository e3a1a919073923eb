mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export BEATAMD_GF_KERNEL=1
for ab in 0 1 2 3; do BEATAMD_GS_ABLATE=$ab timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --chains 256 2>&1 | tail -1 > gpurun_out/ab_$ab.json; done
for rb in 75 248; do BEATAMD_GS_RB=$rb timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --chains 256 2>&1 | tail -1 > gpurun_out/ab_rb$rb.json; done
BEATAMD_GS_CG=128 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --chains 256 2>&1 | tail -1 > gpurun_out/ab_cg128.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value'],1), round(d['kernel_ms_per_step']['gfstack'],3))
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
PY
