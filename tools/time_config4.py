import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, beat_amd
from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
ctx = beat_amd.get_context(0); ctx.use_torch_stream()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
spec = SyntheticSpec((10, 10), (20, 20), (2.0, 2.0), T=35, N=N, D=2, S=60, st_dt=0.5, slip_varnames=("uparr", "uperp"),
                     covariance="toeplitz", station_shifts=True, geodetic_nobs=(214, 214), vel_bounds=(3.0, 4.0),
                     time_bounds=(0.0, 2.0), interpolation=sys.argv[2] if len(sys.argv) > 2 else "multilinear")
prob, host = build_problem(spec, device_library=True, ctx=ctx)
f = prob.compile(ctx)
lay = host["layout"]
Q = torch.from_numpy(draw_population(spec, lay, host["lower"], host["upper"], 512)).cuda()
L = f.batch(Q); torch.cuda.synchronize()
for rep in range(2):
    ctx.enable_timing(True); ctx.reset_timing()
    t0 = time.perf_counter()
    for i in range(20):
        L = f.batch(Q)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    names = ("sweep", "tables", "grouptables", "gfstack", "gfstack_standin", "quadform", "finish", "astep", "geostack")
    print("issue %.3f ms/step, wall %.3f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3), ctx.last_kernel(),
          {k: round(ctx.kernel_time(k)[0] / 20, 3) for k in names if ctx.kernel_time(k)[1]})
    ctx.enable_timing(False)
t0 = time.perf_counter()
for i in range(20):
    L = f.batch(Q)
torch.cuda.synchronize()
print("no timers: wall %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
# the Metropolis step of bench.py's legs (propose -> forward -> accept)
lo, up = lay.bounds(host["lower"], host["upper"])
lo_s, up_s = torch.from_numpy(lo).cuda(), torch.from_numpy(up).cuda()
nsw = 16
delta = torch.randn((nsw, 512, lay.size), device="cuda", dtype=torch.float64) * (5e-4 * (up_s - lo_s))
log_u = torch.log(torch.rand((nsw, 512), device="cuda", dtype=torch.float64))
scaling = torch.ones(512, device="cuda", dtype=torch.float64)
acc = torch.zeros(512, device="cuda", dtype=torch.int32)
L0 = f.batch(Q)
for i in range(nsw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f.astep_batch(Q, L0, delta[i], scaling, lo_s, up_s, log_u[i], 2e-6, acc)
    torch.cuda.synchronize()
    print("astep %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), end="; ")
print()
