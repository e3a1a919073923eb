# geometry-mode step on the GPU: tests, then the per-step time of BASELINE configs[1] (tools/geo_app.py)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_geometry.py tests/test_gpu_samplers.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -5
for v in "BEATAMD_GEOM_WAVES=2" "BEATAMD_GEOM_WAVES=3" "BEATAMD_GEOM_WAVES=4" "BEATAMD_GEOM_OWN=1" "BEATAMD_QS_KB=16" "BEATAMD_GEOM_WAVES=3 BEATAMD_QS_KB=16"; do echo $v; env $v timeout 120 python tools/geo_app.py 1024 200 2>&1 | grep "stage 2\|one batched"; done
