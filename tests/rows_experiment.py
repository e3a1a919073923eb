#!/usr/bin/env python
"""rows_experiment.py -- not a test: the CPU experiment behind the row passes of the stacking kernels (round 5).

For prior draws of (a) the bench model (config 3: D=3 x S=25), (b) a library on the grid of the reference's tutorial
(durations 0-4 s every 0.25 s: D=17; start times 0-20 s every 0.5 s: S=41; docs/examples/FFI_kinematic.rst:185-195)
and (c) BASELINE configs[3] (two subfaults, D=2 x S=60, station shifts) it counts per (chain group, patch)

  * nearest neighbour: the distinct library rows the group touches and the passes of <= 96 rows they make
    (k_gfstack_ws: three LDS row buffers of 96 slots);
  * multilinear: the distinct rows of the group's (duration, start-time) cells, the passes a greedy cut of the cell
    order makes under a budget of row slots, and the chains a consumer wavefront (37 chains) has in a pass.

Start times / indices come from the oracle (test infrastructure, which is why this file lives under tests/).

    python tests/rows_experiment.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beat_amd.synthetic import SyntheticSpec, _layout_and_bounds, draw_population  # noqa: E402
from oracle import oracle as orc  # noqa: E402

NCH, NW = 37, 14


def indices(spec, C, interp, t_of=0):
    """-> (di [C,P], si [C,P], hyp [C,2]) grid indices of target t_of (nn: the node; multilinear: the ceil node)"""
    lay, lo, up = _layout_and_bounds(spec)
    Q = draw_population(spec, lay, lo, up, C)
    P = spec.P
    si = np.zeros((C, P), dtype=int)
    di = np.zeros((C, P), dtype=int)
    hyp = np.zeros((C, 2))
    for c in range(C):
        pt = lay.rmap(Q[c])
        st0 = np.empty(P)
        o = 0
        for k, (nd, ns, h) in enumerate(zip(spec.n_patch_dip, spec.n_patch_strike, spec.patch_size)):
            hd, hs = orc.positions2idxs([pt["nucleation_dip"][k], pt["nucleation_strike"][k]], h)
            n = nd * ns
            st0[o:o + n] = orc.fast_sweep(1.0 / pt["velocities"][o:o + n], h, int(hd), int(hs), nd, ns) + pt["time"][k]
            o += n
        if spec.station_shifts:
            nst = lay.varsizes["time_shifts_any_P_0"]
            st0 = st0 - pt["time_shifts_any_P_0"][t_of % nst]
        si[c], _ = orc.time2idx(st0, spec.st_min, spec.st_dt, interp)
        di[c], _ = orc.time2idx(pt["durations"], spec.du_min, spec.du_dt, interp)
        hyp[c] = (pt["nucleation_dip"][0], pt["nucleation_strike"][0])
    return di, si, hyp


def hyp_order(hyp, C):
    """k_gc_order with caller keys: 4 bands of whole wavefronts by strike, inside a band by dip"""
    r0 = np.argsort(np.argsort(hyp[:, 1], kind="stable"), kind="stable")
    nw = (C + NCH - 1) // NCH
    band = (r0 // NCH) * 4 // nw
    return np.lexsort((np.arange(C), hyp[:, 0], band))


def nn_stats(di, si, S, cap=96):
    C, P = di.shape
    u = np.array([len(np.unique(di[:, p] * S + si[:, p])) for p in range(P)])
    passes = np.ceil(u / float(cap)).astype(int)
    return dict(rows_mean=u.mean(), rows_max=u.max(), passes_mean=passes.mean(), passes_max=passes.max())


def ml_stats(di, si, D, S, order, cap):
    """greedy cut of the cell order (ceil-d, ceil-s ascending) under `cap` row slots"""
    C, P = di.shape
    S1 = S + 1
    rows, npass, nch = [], [], []
    for p in range(P):
        dc, sc = di[:, p], si[:, p]
        df = (dc + D - 1) % D
        cells = np.unique(dc * S1 + sc)
        sub_of = {}
        have, n, k = set(), 0, 0
        allrows = set()
        for cell in cells:
            d, s = divmod(int(cell), S1)
            f = (d + D - 1) % D
            need = {(f, s), (f, s + 1), (d, s), (d, s + 1)}
            allrows |= need
            new = len(need - have)
            if n + new > cap:
                k += 1
                have, n = set(), 0
                new = 4 if f != d else 2
            have |= need
            n = len(have)
            sub_of[int(cell)] = k
        rows.append(len(allrows))
        npass.append(k + 1)
        sub = np.array([sub_of[int(x)] for x in dc * S1 + sc])
        for w in range((C + NCH - 1) // NCH):
            ids = order[w * NCH:(w + 1) * NCH]
            cnt = np.bincount(sub[ids], minlength=k + 1)
            nch.extend(cnt.tolist())
    nch = np.array(nch)
    return dict(rows_mean=np.mean(rows), rows_max=np.max(rows), passes_mean=np.mean(npass), passes_max=np.max(npass),
                chains_per_wave_pass_mean=nch.mean(), chains_per_wave_pass_p90=np.percentile(nch, 90),
                chains_per_wave_pass_max=nch.max())


def main():
    C = 512
    cases = [
        ("config 3 (D=3, S=25)", dict(n_patch_dip=(20,), n_patch_strike=(20,), patch_size=(1.0,), D=3, S=25,
                                      nuc_margin=0.0, time_bounds=(0.0, 0.0))),
        ("tutorial grid (D=17 @0.25 s, S=41 @0.5 s)", dict(n_patch_dip=(20,), n_patch_strike=(20,), patch_size=(1.0,),
                                                          D=17, S=41, du_min=0.0, du_dt=0.25, nuc_margin=0.0,
                                                          time_bounds=(0.0, 0.0))),
        ("configs[3] (2 subfaults, D=2, S=60, station shifts)",
         dict(n_patch_dip=(10, 10), n_patch_strike=(20, 20), patch_size=(2.0, 2.0), D=2, S=60, st_dt=0.5,
              slip_varnames=("uparr", "uperp"), station_shifts=True, vel_bounds=(3.0, 4.0), time_bounds=(0.0, 2.0), T=35)),
    ]
    for name, kw in cases:
        print("== %s, %d chains" % (name, C))
        for interp in ("nearest_neighbor", "multilinear"):
            spec = SyntheticSpec(N=64, interpolation=interp, **{"T": 4, **kw})
            di, si, hyp = indices(spec, C, interp)
            if interp == "nearest_neighbor":
                print("   nn :", {k: round(float(v), 2) for k, v in nn_stats(di, si, spec.S).items()})
            else:
                order = hyp_order(hyp, C)
                for cap in (100, 156):
                    print("   ml  cap %3d:" % cap,
                          {k: round(float(v), 2) for k, v in ml_stats(di, si, spec.D, spec.S, order, cap).items()})


if __name__ == "__main__":
    main()
