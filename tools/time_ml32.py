"""time the multilinear stacking at the bench shape with float-stored libraries (k_gfstack_dmaf) against the
float64 cell kernel: python tools/time_ml32.py [chains=512]"""
import os, sys
os.environ.setdefault("BEATAMD_KNOBS_LIVE", "1")   # this tool flips the BEATAMD_G* knobs between launches
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_amd
from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = beat_amd.get_context(0)
ctx.use_torch_stream()
for interp in ("multilinear", "nearest_neighbor"):
    spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, interpolation=interp, nuc_margin=0.0,
                         time_bounds=(0.0, 0.0))
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    f = prob.compile(ctx)
    Q = torch.from_numpy(draw_population(spec, host["layout"], host["lower"], host["upper"], C)).cuda()
    for f32 in (None, True, False, True):   # None: the unrounded library; then float / float64 kernels on the rounded one
        if f32 is True and not getattr(prob.wavemaps[0].gfs["uparr"], "rounded_to_f32", False):
            f.round_libraries_to_f32()
        elif f32 is not None:
            f.set_f32(f32)
        L = f.batch(Q)
        ctx.synchronize()
        ctx.enable_timing(True)
        ctx.reset_timing()
        for _ in range(6):
            L = f.batch(Q)
        ctx.synchronize()
        ms, n = ctx.kernel_time("gfstack")
        ctx.enable_timing(False)
        print("TIME %s f32=%s: %.3f ms per launch  %s  like[0] %.6f" % (interp, f32, ms / n, ctx.last_kernel(), float(L[0, -1])))
    del f, prob
    torch.cuda.empty_cache()
