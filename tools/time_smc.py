"""time_smc.py -- where a Metropolis step of smc_sample goes: config 3, nn, `chains` chains, 2 tempering
stages of `steps` steps, with / without stage files (async / in line): wall per step of the sampling part and the HIP-event
sums of every timer of the context per step.

    python tools/time_smc.py [chains=4096] [steps=30] [modes=none,async,sync]
"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import beat_amd  # noqa: E402
from beat_amd.sampler import SMC, smc_sample  # noqa: E402
from beat_amd.synthetic import SyntheticSpec, build_problem  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
modes = (sys.argv[3] if len(sys.argv) > 3 else "none,async,sync").split(",")
interp = sys.argv[4] if len(sys.argv) > 4 else "nearest_neighbor"
ctx = beat_amd.get_context(0)
ctx.use_torch_stream()
spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, nuc_margin=0.0, time_bounds=(0.0, 0.0), interpolation=interp)
prob, host = build_problem(spec, device_library=True, ctx=ctx)
f = prob.compile(ctx)
lay = host["layout"]
lo, up = lay.bounds(host["lower"], host["upper"])
dev = torch.device("cuda", 0)
KEYS = ("sweep", "tables", "grouptables", "gfstack", "quadform", "finish", "astep", "proposal", "stage")
for mode in modes:
    for rep in range(2):
        st = SMC(f, lo, up, n_chains=C, device=dev, random_seed=11, tune_interval=25)
        home = tempfile.mkdtemp(prefix="beatamd_smc_") if mode != "none" else None
        ctx.enable_timing(True)
        ctx.reset_timing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smc_sample(steps, st, max_stages=2, homepath=home, final_stage=False, layout=lay if home else None,
                   out_names=prob.out_names if home else None, async_stage_files=(mode != "sync"))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        tm = dict(st.timings)
        n = tm.pop("steps")
        kt = {k: round(ctx.kernel_time(k)[0] / n, 3) for k in KEYS if ctx.kernel_time(k)[1]}
        ctx.enable_timing(False)
        if home:
            shutil.rmtree(home, ignore_errors=True)
        print("%d chains %-6s rep %d: wall %.2f s, sampling %.2f ms/step (%.0f chain-steps/s), split %s | event ms/step %s sum %.2f"
              % (C, mode, rep, wall, tm["sample_s"] / n * 1e3, C * n / tm["sample_s"],
                 {k: round(v, 3) for k, v in tm.items()}, kt, sum(kt.values())), flush=True)
        del st
