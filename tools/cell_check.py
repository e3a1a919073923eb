#!/usr/bin/env python
"""cell_check.py -- quick GPU check of k_gfstack_cell (gfcell.hip): bitwise against the streaming
kernel for explicit start times (tables per target) and through the fused model (tables per
patch), all three epilogues.  Development aid; the assertions live in tests/."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import beat_amd  # noqa: E402
from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig  # noqa: E402


def make_lib(ctx, G, st_min, st_dt, du_min, du_dt):
    T, P, D, S, N = G.shape
    cfg = SeismicGFLibraryConfig(dimensions=(T, P, D, S, N), starttime_sampling=st_dt,
                                 duration_sampling=du_dt, starttime_min=st_min, duration_min=du_min)
    gf = SeismicGFLibrary(cfg)
    gf.setup(T, P, D, S, N, allocate=True)
    gf._gfmatrix[:] = G
    gf.init_optimization(ctx)
    return gf


def main():
    ctx = beat_amd.get_context(0)
    ok = True
    for (T, P, D, S, N, C) in ((2, 5, 3, 9, 64, 512), (3, 17, 3, 25, 200, 530), (2, 6, 2, 5, 130, 1100),
                               (1, 3, 3, 6, 62, 200)):
        rng = np.random.default_rng(C + N)
        G = rng.standard_normal((T, P, D, S, N))
        gf = make_lib(ctx, G, 0.0, 0.5, 0.5, 0.5)
        dur = rng.uniform(0.5, 0.5 + 0.5 * (D - 1), (C, P))
        st = rng.uniform(0.0, 0.5 * (S - 1) - 0.01, (C, T, P))
        sl = rng.uniform(0, 5, (C, P))
        os.environ["BEATAMD_GF_KERNEL"] = "0"
        a = gf.stack_all_batch(dur, st, sl, interpolation="multilinear")
        ka = ctx.last_kernel()
        del os.environ["BEATAMD_GF_KERNEL"]
        for srt, ml, runs in (("1", "0", "0"), ("0", "0", "0"), ("0", "1", "0"), ("1", "1", "1"), ("0", "1", "1")):
            os.environ["BEATAMD_GC_SORT"] = srt
            os.environ["BEATAMD_GS_CELL"] = "1"
            os.environ["BEATAMD_GS_ML"] = ml
            os.environ["BEATAMD_GS_RUNS"] = runs
            b = gf.stack_all_batch(dur, st, sl, interpolation="multilinear")
            kb = ctx.last_kernel()
            eq = np.array_equal(a, b)
            print("stack", (T, P, D, S, N, C), "sort", srt, ka, kb, "bitwise", eq,
                  "maxdiff", float(np.abs(a - b).max()), flush=True)
            ok &= eq and kb.startswith(("k_gfstack_runs" if runs == "1" else "k_gfstack_ml") if ml == "1" else "k_gfstack_cell")
    # fused model: tables per patch, scalar and dense covariance
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    for cov in ("scalar", "toeplitz"):
        spec = SyntheticSpec((6,), (6,), (1.0,), T=6, N=256, D=3, S=25, covariance=cov,
                             interpolation="multilinear")
        prob, host = build_problem(spec)
        f = prob.compile(ctx)
        Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 700)
        os.environ["BEATAMD_GF_KERNEL"] = "0"
        A = f.batch(Q)
        ka = ctx.last_kernel()
        del os.environ["BEATAMD_GF_KERNEL"]
        for ml, runs in (("0", "0"), ("1", "0"), ("1", "1")):
            os.environ["BEATAMD_GS_CELL"] = "1"
            os.environ["BEATAMD_GS_ML"] = ml
            os.environ["BEATAMD_GS_RUNS"] = runs
            B = f.batch(Q)
            kb = ctx.last_kernel()
            d = float(np.abs(np.asarray(A) - np.asarray(B)).max() / np.abs(np.asarray(A)).max())
            print("model", cov, ka, kb, "rel diff", d, flush=True)
            ok &= d < 1e-12 and kb.startswith(("k_gfstack_runs" if runs == "1" else "k_gfstack_ml") if ml == "1" else "k_gfstack_cell")
    print("CELL_CHECK", "OK" if ok else "FAILED", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
