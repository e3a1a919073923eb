set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-r2d}
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -rf -k "${PYTEST_K:-shared_row or window or fullsize_shipped or fused_model or fullsize_fused}" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 900 python tools/exp_variants.py $O/variants.jsonl ${VARIANTS:-tools/variants_c.json} > $O/variants.log 2>&1
tail -2 $O/variants.log
