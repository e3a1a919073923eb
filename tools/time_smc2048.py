"""time_smc2048.py -- smc_sample at 2048 chains on the config-3 problem with and without the hypocentre cut of
multi-group batches (BEATAMD_GC_GLOBAL), per-kernel timers: python tools/time_smc2048.py"""
import os, sys, time
os.environ.setdefault("BEATAMD_KNOBS_LIVE", "1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, beat_amd
from beat_amd.sampler import SMC, smc_sample
from beat_amd.synthetic import SyntheticSpec, build_problem
ctx = beat_amd.get_context(0); ctx.use_torch_stream()
spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=int(sys.argv[1]) if len(sys.argv) > 1 else 4096, D=3, S=25, nuc_margin=0.0, time_bounds=(0.0, 0.0))
prob, host = build_problem(spec, device_library=True, ctx=ctx)
f = prob.compile(ctx)
lo, up = host["layout"].bounds(host["lower"], host["upper"])
for knob in ("1", "0", "1", "0"):
    os.environ["BEATAMD_GC_GLOBAL"] = knob
    st = SMC(f, lo, up, n_chains=2048, device=torch.device("cuda", 0), random_seed=11, tune_interval=25)
    ctx.enable_timing(True); ctx.reset_timing()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pop, lp, betas = smc_sample(50, st, max_stages=3, final_stage=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    names = ("sweep", "tables", "grouptables", "gfstack", "finish", "astep", "proposal")
    print("GC_GLOBAL", knob, "wall %.2f s" % dt, "%.0f chain-steps/s" % (2048 * st.timings["steps"] / dt), ctx.last_kernel(),
          {k: (round(ctx.kernel_time(k)[0] / max(ctx.kernel_time(k)[1], 1), 3), ctx.kernel_time(k)[1]) for k in names}, ctx.gf_group_stats(), flush=True)
    ctx.enable_timing(False)
