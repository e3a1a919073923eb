#!/usr/bin/env python
"""order_experiment.py -- not a test: the CPU experiment behind the chain order of k_gfstack_runs (k_gc_order in
beat_amd/csrc/gfcell.hip).  For 512 prior draws of the bench model (20 x 20 patches, multilinear library 3 x 25)
it counts, per patch, the distinct (duration, start-time) cells the 14 wavefronts of a 512-chain group have to read
under different chain orders.  Start times / indices come from the oracle (test infrastructure, which is why this
file lives under tests/).

    python tests/order_experiment.py            (one 512-chain group: the order inside a group)
    python tests/order_experiment.py 2048       (several groups: how the batch is cut into its groups, k_gc_cut)

round 5, 2048 chains = 4 groups of 518 (new cells per patch summed over the 56 wavefronts / q / distinct cells per group):
    groups as the chains come, 4 bands by strike then dip       485   q 4.23   27.8
    4 strips along strike (round 5, first version), 4 bands     392   q 5.23   19.2
    4 strips along strike, bands from the group's extents (2)   327   q 6.27   19.2
    bisection along the wider key = 2 x 2 quadrants, 4 bands    328   q 6.24   16.9   <- k_gc_cut + k_gc_order
    (a third key -- the chain's mean start index, i.e. its rupture speed -- does not help: q 6.2 -> 6.1)

round 4 (q = chains per cell read):
    chains as they come                                    252 cells / patch   q 2.03
    round-3 order (8 bands by s[0], then s[P/2])           160                 q 3.21
    5 bands of whole wavefronts by s[0], then s[P/2]       140                 q 3.66   <- without caller keys
    4 bands of whole wavefronts by hypocentre strike, dip  121                 q 4.2    <- with GfStackCall::order_key
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beat_amd.synthetic import SyntheticSpec, _layout_and_bounds, draw_population  # noqa: E402
from oracle import oracle as orc  # noqa: E402

C, P, NCH, NW = 512, 400, 37, 14
CG = NCH * NW


def batch_cut(Cb):
    """Cb chains in groups of 518: new cells per patch / distinct cells per group under different cuts of the batch"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gfcell_emu as emu
    spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, interpolation="multilinear",
                         nuc_margin=0.0, time_bounds=(0.0, 0.0))
    lay, lo, up = _layout_and_bounds(spec)
    Q = draw_population(spec, lay, lo, up, Cb)
    key = np.zeros((Cb, P), dtype=int)
    hyp = np.zeros((Cb, 2))
    for c in range(Cb):
        pt = lay.rmap(Q[c])
        hd, hs = orc.positions2idxs([pt["nucleation_dip"][0], pt["nucleation_strike"][0]], 1.0)
        st0 = orc.fast_sweep(1.0 / pt["velocities"], 1.0, int(hd), int(hs), 20, 20) + pt["time"][0]
        key[c] = orc.time2idx(pt["durations"], 0.5, 0.5, "multilinear")[0] * 64 + orc.time2idx(st0, 0.0, 0.5, "multilinear")[0]
        hyp[c] = (pt["nucleation_strike"][0], pt["nucleation_dip"][0])
    ng = (Cb + CG - 1) // CG

    def in_group(ids, nb):
        n = len(ids)
        nw = (n + NCH - 1) // NCH
        if nb is None:      # k_gc_order: bands from the group's extents
            e0, e1 = np.ptp(hyp[ids, 0]), np.ptp(hyp[ids, 1])
            nb = max(1, min(nw, int(np.rint(np.sqrt(nw * e0 / e1)))))
        r0 = np.argsort(np.argsort(hyp[ids, 0], kind="stable"), kind="stable")
        return ids[np.lexsort((np.arange(n), hyp[ids, 1], (r0 // NCH) * nb // nw))]

    def report(name, groups, nb):
        new = dist = 0
        for g in groups:
            o = in_group(g, nb)
            for w in range(0, len(o), NCH):
                new += sum(len(np.unique(key[o[w:w + NCH], p])) for p in range(P))
            dist += sum(len(np.unique(key[g, p])) for p in range(P))
        print("%-62s new cells / patch %6.1f   q = %.2f   distinct cells per group %.1f" % (name, new / P, Cb / (new / P), dist / P / len(groups)))

    ids = np.arange(Cb)
    o = np.argsort(hyp[:, 0], kind="stable")
    m = emu.gc_cut(hyp[:, 0], hyp[:, 1], Cb, CG, ng)
    report("groups as the chains come, 4 bands", [ids[g * CG:(g + 1) * CG] for g in range(ng)], 4)
    report("strips along strike, 4 bands", [o[g * CG:(g + 1) * CG] for g in range(ng)], 4)
    report("strips along strike, bands from the extents", [o[g * CG:(g + 1) * CG] for g in range(ng)], None)
    report("bisection along the wider key (k_gc_cut), 4 bands", [m[g * CG:(g + 1) * CG] for g in range(ng)], 4)
    report("bisection along the wider key, bands from the extents", [m[g * CG:(g + 1) * CG] for g in range(ng)], None)


def main():
    spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, interpolation="multilinear",
                         nuc_margin=0.0, time_bounds=(0.0, 0.0))
    lay, lo, up = _layout_and_bounds(spec)
    Q = draw_population(spec, lay, lo, up, C)
    sc = np.zeros((C, P), dtype=int)
    dc = np.zeros((C, P), dtype=int)
    hyp = np.zeros((C, 2))
    for c in range(C):
        pt = lay.rmap(Q[c])
        hd, hs = orc.positions2idxs([pt["nucleation_dip"][0], pt["nucleation_strike"][0]], 1.0)
        st0 = orc.fast_sweep(1.0 / pt["velocities"], 1.0, int(hd), int(hs), 20, 20) + pt["time"][0]
        sc[c], _ = orc.time2idx(st0, 0.0, 0.5, "multilinear")
        dc[c], _ = orc.time2idx(pt["durations"], 0.5, 0.5, "multilinear")
        hyp[c] = (pt["nucleation_dip"][0], pt["nucleation_strike"][0])
    key = dc * 64 + sc

    def cells(order):
        tot = 0
        for w in range(NW):
            ids = order[w * NCH:(w + 1) * NCH]
            ids = ids[ids < C]
            if len(ids):
                tot += sum(len(np.unique(key[ids, p])) for p in range(P))
        return tot / P

    def padded(o):
        x = np.full(NW * NCH, 10 ** 6)
        x[:C] = o
        return x

    def banded(k1, nb, k2, whole_waves):
        r0 = np.argsort(np.argsort(k1, kind="stable"), kind="stable")
        band = (r0 // NCH) * nb // NW if whole_waves else r0 * nb // C
        return padded(np.lexsort((np.arange(C), k2, band)))

    rows = [("chains as they come", padded(np.arange(C))),
            ("8 bands by s[0], then s[P/2] (round 3)", banded(sc[:, 0], 8, sc[:, P // 2], False)),
            ("5 bands of whole wavefronts by s[0], then s[P/2]", banded(sc[:, 0], 5, sc[:, P // 2], True)),
            ("4 bands of whole wavefronts by hypocentre strike, then dip", banded(hyp[:, 1], 4, hyp[:, 0], True))]
    for name, o in rows:
        n = cells(o)
        print("%-62s %6.1f cells per patch   q = %.2f" % (name, n, C / n))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        batch_cut(int(sys.argv[1]))
    else:
        main()
