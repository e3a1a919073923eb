#!/usr/bin/env python
"""order_experiment.py -- not a test: the CPU experiment behind the chain order of k_gfstack_runs (k_gc_order in
beat_amd/csrc/gfcell.hip).  For 512 prior draws of the bench model (20 x 20 patches, multilinear library 3 x 25)
it counts, per patch, the distinct (duration, start-time) cells the 14 wavefronts of a 512-chain group have to read
under different chain orders.  Start times / indices come from the oracle (test infrastructure, which is why this
file lives under tests/).

    python tests/order_experiment.py

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
    main()
