#!/bin/bash
# repeat the bitwise parity tests of the stacking kernels (race detection) and the full-size tests
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x -k "shared or fused or selection or degenerate" 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -1
# determinism of the default bench path: two evaluations of the same population must be bit-identical
timeout 300 python - <<'PY'
import numpy as np, torch, beat_amd
from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
ctx = beat_amd.get_context(0)
spec = SyntheticSpec((20,), (20,), (1.0,), T=16, N=4096, D=3, S=25, nuc_margin=6.0, time_bounds=(0.0, 0.5))
prob, host = build_problem(spec, device_library=True, ctx=ctx)
f = prob.compile(ctx)
lay = host["layout"]
Q = torch.from_numpy(draw_population(spec, lay, host["lower"], host["upper"], 512)).cuda()
ref = f.batch(Q).cpu().numpy()
bad = 0
for i in range(20):
    out = f.batch(Q).cpu().numpy()
    bad += int(not np.array_equal(out, ref))
print("repeat evaluations differing from the first:", bad, "of 20")
PY
