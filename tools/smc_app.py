"""Application-level check: a few SMC stages on the config-3 problem (62.9 GB library), all
steps through beat_amd.sampler.SMC (device proposals + fused astep), timing per chain-step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import beat_amd  # noqa: E402
from beat_amd.sampler import SMC  # noqa: E402
from beat_amd.synthetic import SyntheticSpec, build_problem  # noqa: E402

n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = beat_amd.get_context(0)
spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, nuc_margin=6.0,
                     time_bounds=(0.0, 0.5))
prob, host = build_problem(spec, device_library=True, ctx=ctx)
f = prob.compile(ctx)
lay = host["layout"]
lo, up = lay.bounds(host["lower"], host["upper"])
dev = torch.device("cuda", 0)
step = SMC(f, lo, up, n_chains=n_chains, tune_interval=10, device=dev, random_seed=1)
step.initialize_population()
Q = step._local(step.array_population)
L = step.stepper.evaluate(Q)
step.select_end_points(Q, L)
for stage in range(3):
    t0 = time.perf_counter()
    step.beta, step.old_beta, step.weights = step.calc_beta()
    step.covariance = step.calc_covariance(repair=False)
    step.set_stage_proposal()
    step.resampling_indexes = step.resample()
    step.stage += 1
    t1 = time.perf_counter()
    Q, L = step.sample_stage(n_steps)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    step.select_end_points(Q, L)
    t3 = time.perf_counter()
    print("stage %d beta %.3e: transition %.3f s, sampling %.3f s = %.1f chain-steps/s "
          "(%.2f ms/step), gather %.3f s, acceptance %.3f, mean like %.4e"
          % (step.stage, step.beta, t1 - t0, t2 - t1, n_chains * n_steps / (t2 - t1),
             (t2 - t1) / n_steps * 1e3, t3 - t2, step.stage_acceptance[-1], step.likelihoods.mean()))
