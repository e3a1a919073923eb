set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -rf -x -k "shared_row or window or 512_chain" > $O/pytest_ws.log 2>&1
tail -5 $O/pytest_ws.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -rf -x -k "shipped or fused_logp" > $O/pytest_full.log 2>&1
tail -5 $O/pytest_full.log
timeout 900 python tools/exp_variants.py $O/variants_ws.jsonl tools/variants_ws.json > $O/variants_ws.log 2>&1
cat $O/variants_ws.jsonl | cut -c1-400
