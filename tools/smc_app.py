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
spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, time_bounds=(0.0, 0.0))
prob, host = build_problem(spec, device_library=True, ctx=ctx)
f = prob.compile(ctx)
lay = host["layout"]
lo, up = lay.bounds(host["lower"], host["upper"])
dev = torch.device("cuda", 0)
step = SMC(f, lo, up, n_chains=n_chains, tune_interval=10, device=dev, random_seed=1)
Q = step.initialize_population()
L = step.stepper.evaluate(Q)
step.select_end_points(Q, L)
for stage in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.transition()          # weights, beta, proposal factor, resampling: on the device
    step.stage += 1
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    Q, L = step.sample_stage(n_steps)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    step.select_end_points(Q, L)
    t3 = time.perf_counter()
    print("stage %d beta %.3e: transition %.2f ms, sampling %.3f s = %.1f chain-steps/s "
          "(%.2f ms/step), gather %.2f ms, acceptance %.3f, mean like %.4e"
          % (step.stage, step.beta, (t1 - t0) * 1e3, t2 - t1, n_chains * n_steps / (t2 - t1),
             (t2 - t1) / n_steps * 1e3, (t3 - t2) * 1e3, step.stage_acceptance[-1],
             float(step.likelihoods.mean())))
