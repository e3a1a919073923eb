"""BASELINE configs[1] shape: rectangular-source geodetic composite (two SAR scenes, 214 + 205
points, full covariances), 1024 SMC chains on one GPU -- chain-steps/s of the sampling stage and
the time of one batched likelihood evaluation (geometry.hip k_geom_los + residual + MVN)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import beat_amd  # noqa: E402
from beat_amd.sampler import SMC  # noqa: E402
from test_geometry import _problem  # noqa: E402  (the test's problem builder: Laquila geometry)

n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
use_graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
ctx = beat_amd.get_context(0)
rng = np.random.default_rng(11)
prob, lay, lower, upper = _problem(rng, (214, 205))
lo, up = lay.bounds(lower, upper)
f = prob.compile(ctx)
dev = torch.device("cuda", 0)
step = SMC(f, lo, up, n_chains=n_chains, tune_interval=10, device=dev, random_seed=2, use_graph=use_graph)
Q = step.initialize_population()
L = step.stepper.evaluate(Q)
step.select_end_points(Q, L)
for stage in range(2):
    step.transition()
    step.stage += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Q, L = step.sample_stage(n_steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step.select_end_points(Q, L)
    print("stage %d: %d chains x %d steps in %.3f s = %.0f chain-steps/s (%.1f us per step of the whole population)"
          % (step.stage, n_chains, n_steps, dt, n_chains * n_steps / dt, dt / n_steps * 1e6))
Qd = Q.contiguous()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    LL = f.batch(Qd)
torch.cuda.synchronize()
print("one batched likelihood of %d chains: %.1f us" % (n_chains, (time.perf_counter() - t0) / 200 * 1e6))
