"""Application-level check of the parallel-tempering path on the config-3 problem with a dense
Toeplitz data covariance (BASELINE configs[4] is 32 temperatures x 256 replicas over 8 GPUs = 1024
chains per GPU): `pt_sample` end to end on one GPU -- device proposals, the fused tempered astep,
exchange rounds (likelihood all-gather + device permutation) -- timing per chain-step.

    python tools/pt_app.py [temperatures=4] [replicas=256] [samples=2048] [covariance=toeplitz]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import beat_amd  # noqa: E402
from beat_amd.sampler import pt_sample  # noqa: E402
from beat_amd.synthetic import SyntheticSpec, build_problem  # noqa: E402

n_temp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_samples = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
cov = sys.argv[4] if len(sys.argv) > 4 else "toeplitz"
ctx = beat_amd.get_context(0)
spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, time_bounds=(0.0, 0.0), covariance=cov)
prob, host = build_problem(spec, device_library=True, ctx=ctx)
f = prob.compile(ctx)
lo, up = host["layout"].bounds(host["lower"], host["upper"])
kw = dict(n_chains_posterior=1, n_chains_tempered=n_temp - 1, n_replicas=n_rep, swap_interval=(3, 5),
          beta_tune_interval=4, proposal_cov=np.diag(((up - lo) * 5e-4) ** 2),
          device=torch.device("cuda", 0), random_seed=5)
pt_sample(f, lo, up, n_samples=n_rep, **kw)   # warm-up: allocations, the measured group size
ctx.enable_timing(True)
ctx.reset_timing()
torch.cuda.synchronize()
t0 = time.perf_counter()
s, ls, man = pt_sample(f, lo, up, n_samples=n_samples, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n_chains = man.n_workers * n_rep
gf_ms, n_launch = ctx.kernel_time("gfstack")
qf_ms, n_qf = ctx.kernel_time("quadform")
ctx.enable_timing(False)
print("PT %d temperatures x %d replicas = %d chains, covariance %s: %d posterior samples in %.2f s"
      % (man.n_workers, n_rep, n_chains, cov, s.shape[0], dt))
print("  %d forward launches of %d chains -> %.0f chain-steps/s end to end (%.2f ms per launch incl. exchange rounds);"
      " stacking %.2f ms, quadform %.2f ms per launch; %d exchange rounds, last kernel %s"
      % (n_launch, n_chains, n_launch * n_chains / dt, dt / max(n_launch, 1) * 1e3,
         gf_ms / max(n_launch, 1), qf_ms / max(n_qf, 1), len(man.history), ctx.last_kernel()))
assert np.isfinite(ls).all()
