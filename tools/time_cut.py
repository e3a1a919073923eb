"""time_cut.py -- how a batch of several chain groups is cut into its groups (k_gc_cut, gfcell.hip), A/B in one session:
BEATAMD_GC_GLOBAL = 1 (bisection along the hypocentre key of the wider extent: pieces of the fault), 2 (strips in the
order of the first key: round 5's first version), 0 (groups as the chains come) and, for the multilinear kernel, the
bands of whole wavefronts inside a group (BEATAMD_GC_BANDS: 0 = from the group's extents, 4 = round 4's fixed four).
Config 3 (D = 3, S = 25, N = 4096) and the tutorial grid (D = 17, S = 41, N = 512), nn and multilinear.

    python tools/time_cut.py [chains=2048] [steps=6]
"""
import os
import sys
import time

os.environ.setdefault("BEATAMD_KNOBS_LIVE", "1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

import beat_amd  # noqa: E402
from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = beat_amd.get_context(0)
ctx.use_torch_stream()


def leg(name, spec, variants):
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    f = prob.compile(ctx)
    Q = torch.from_numpy(draw_population(spec, host["layout"], host["lower"], host["upper"], C)).cuda()
    ref = None
    for rep in range(2):
        for env in variants:
            for k in ("BEATAMD_GC_GLOBAL", "BEATAMD_GC_BANDS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            L = f.batch(Q)
            torch.cuda.synchronize()
            if ref is None:
                ref = L.clone()
            same = bool(torch.equal(L, ref))
            ctx.enable_timing(True)
            ctx.reset_timing()
            t0 = time.perf_counter()
            for _ in range(steps):
                L = f.batch(Q)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps * 1e3
            kt = {k: round(ctx.kernel_time(k)[0] / steps, 3) for k in ("gfstack", "grouptables", "tables") if ctx.kernel_time(k)[1]}
            ctx.enable_timing(False)
            st = ctx.gf_group_stats()
            pl = ctx.gf_plan()
            print("%-28s %-40s wall %.2f ms/step %s rows/group-patch %.1f passes %.2f bitwise %s %s" %
                  (name, env, wall, kt, st["mean_rows"], pl["mean_passes"], same, ctx.last_kernel()), flush=True)
    f.release()
    del f, prob, host, Q
    import gc
    gc.collect()
    torch.cuda.empty_cache()


cut = [{}, {"BEATAMD_GC_GLOBAL": "2"}, {"BEATAMD_GC_GLOBAL": "0"}]
cut_ml = [{}, {"BEATAMD_GC_BANDS": "4"}, {"BEATAMD_GC_GLOBAL": "2", "BEATAMD_GC_BANDS": "4"}, {"BEATAMD_GC_GLOBAL": "2"}, {"BEATAMD_GC_GLOBAL": "0"}]
which = sys.argv[3] if len(sys.argv) > 3 else "all"
c3 = dict(T=64, N=4096, D=3, S=25, nuc_margin=0.0, time_bounds=(0.0, 0.0))
grid = dict(T=64, N=512, D=17, S=41, st_min=0.0, st_dt=0.5, du_min=0.0, du_dt=0.25, nuc_margin=0.0, time_bounds=(0.0, 0.0))
if which in ("all", "c3"):
    leg("config 3 nn", SyntheticSpec((20,), (20,), (1.0,), **c3), cut)
    leg("config 3 multilinear", SyntheticSpec((20,), (20,), (1.0,), interpolation="multilinear", **c3), cut_ml)
if which in ("all", "grid"):
    leg("tutorial grid nn", SyntheticSpec((20,), (20,), (1.0,), **grid), cut)
    leg("tutorial grid multilinear", SyntheticSpec((20,), (20,), (1.0,), interpolation="multilinear", **grid), cut_ml)
