#!/usr/bin/env python
"""time_ml.py -- time the stacking kernel of the fused FFI model at the bench shape (development
aid): python tools/time_ml.py [--interp multilinear] [--chains 512] [--reps 6]; prints the
average launch time of the 'gfstack' timer and the kernel that ran."""
import argparse
import os

os.environ.setdefault("BEATAMD_KNOBS_LIVE", "1")   # this tool flips the BEATAMD_G* knobs between launches
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--interp", default="multilinear")
    ap.add_argument("--chains", type=int, default=512)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--targets", type=int, default=64)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--sort-chains", action="store_true", help="order the population by hypocentre (strike bands, then dip) first")
    ap.add_argument("--envs", default="", help="semicolon separated VAR=VALUE sets (comma separated inside a set) to time in turn")
    args = ap.parse_args()
    import torch
    import beat_amd
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    ctx = beat_amd.get_context(0)
    ctx.use_torch_stream()
    spec = SyntheticSpec((20,), (20,), (1.0,), T=args.targets, N=args.samples, D=3, S=25,
                         interpolation=args.interp, nuc_margin=0.0, time_bounds=(0.0, 0.0))
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    f = prob.compile(ctx)
    Q = torch.from_numpy(draw_population(spec, host["layout"], host["lower"], host["upper"], args.chains)).to("cuda")
    if args.sort_chains:
        lay = host["layout"]
        qs = Q.cpu().numpy()
        ks = qs[:, lay.offset("nucleation_strike")]
        kd = qs[:, lay.offset("nucleation_dip")]
        band = np.argsort(np.argsort(ks)) * 8 // len(ks)
        Q = Q[torch.from_numpy(np.lexsort((kd, band))).to(Q.device)].contiguous()
    sets = [s for s in args.envs.split(";")] if args.envs else [""]
    ref = None
    for es in sets:
        kv = dict(x.split("=") for x in es.split(",") if x)
        for k, v in kv.items():
            os.environ[k] = v
        L = f.batch(Q)
        ctx.synchronize()
        ctx.enable_timing(True)
        ctx.reset_timing()
        for _ in range(args.reps):
            L = f.batch(Q)
        ctx.synchronize()
        ms, n = ctx.kernel_time("gfstack")
        gms, gn = ctx.kernel_time("grouptables")
        ctx.enable_timing(False)
        like = L[:, -1].cpu().numpy()
        if ref is None:
            ref = like
        dev = float(np.nanmax(np.abs(like - ref) / np.abs(ref))) if np.isfinite(like).all() else float("nan")
        print("TIME %-40s gfstack %.3f ms  tables %.3f ms  %s  like rel dev vs first %.2e"
              % (es or "(default)", ms / max(n, 1), gms / max(gn, 1), ctx.last_kernel(), dev), flush=True)
        for k in kv:
            os.environ.pop(k, None)


if __name__ == "__main__":
    main()
