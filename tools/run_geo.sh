# geometry-mode step on the GPU: tests, then the per-step time of BASELINE configs[1] (tools/geo_app.py)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_geometry.py tests/test_gpu_samplers.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -5
for w in 2 3 4; do echo waves $w; BEATAMD_GEOM_WAVES=$w timeout 120 python tools/geo_app.py 1024 200 2>&1 | grep -v amdgpu.ids; done
