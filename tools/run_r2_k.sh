set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2k
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -rf -x -k "more_than_64 or 512_chain" > $O/pytest_ws.log 2>&1
tail -5 $O/pytest_ws.log
timeout 900 python tools/exp_variants.py $O/variants.jsonl ${VARIANTS:-tools/variants_ws.json} > $O/variants.log 2>&1
python - <<PY
import json
for l in open("$O/variants.jsonl"):
    d = json.loads(l)
    print("%-22s %-24s gfstack %.3f ms  step %.3f ms  %.0f /s" % (d["name"], d.get("kernel"), d.get("gfstack_ms", 0), d.get("ms_per_step", 0), d.get("chain_steps_per_s", 0)))
PY
