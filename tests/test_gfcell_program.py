"""CPU check of the k_gfstack_runs wavefront programs (beat_amd/csrc/gfruns_asm.inc).

tools/gfcell_emu.py interprets the instruction lists that tools/gen_gfruns_asm.py emits -- all 16 wavefronts of a
workgroup with their barriers, the ring of LDS row buffers, the VGPR index register, the scalar descriptor loads --
and a numpy twin of the table builders (chain order, row passes, request / descriptor / weight tables); the result
is compared with a direct multilinear stack (reference beat/ffi/base.py:663-704).  Timing / hazards are not
modelled; the -m gpu tests run the real kernel against k_gfstack and the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_gfruns_asm as gen  # noqa: E402
import gfcell_emu as emu  # noqa: E402


def _reference(G, ro, fa, sl, T, P, N, Gx=(), slx=()):
    """acc = a*b + c in the kernel's order: patches ascending, slip variables, rows k = 0..3"""
    C = sl.shape[0]
    Grs = [g.reshape(-1, N) for g in (G,) + tuple(Gx)]
    sls = (sl,) + tuple(slx)
    out = np.zeros((C, T, N))
    for c in range(C):
        for t in range(T):
            acc = np.zeros(N)
            for p in range(P):
                for Gr, s_ in zip(Grs, sls):
                    for k in range(4):
                        acc = Gr[ro[c, t, p, k]] * (fa[c, t, p, k] * s_[c, p]) + acc
            out[c, t] = acc
    return out


def _run(T, P, D, S, N, C, Ttab_is_one, mode, sort, nth, seed, below_grid=False, nvar=1, cap=None, du_span=None):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((T, P, D, S, N))
    Gx = [rng.standard_normal((T, P, D, S, N)) for _ in range(nvar - 1)]     # libraries of further slip variables
    slx = [rng.uniform(-1, 1, (C, P)) for _ in range(nvar - 1)]
    du = rng.uniform(0.5, 0.5 + 0.5 * (D - 1), (C, P))
    st = rng.uniform(0.0, max(0.5 * (S - 1) - 0.01, 0.0), (C, 1 if Ttab_is_one else T, P))
    st[0, 0, 0] = 0.0        # exactly on node 0: the floor node wraps to the last one with factor 0
    du[min(1, C - 1), P - 1] = 0.5
    if below_grid:
        # times below the first grid node: the reference's floor node wraps to the LAST node with a non-zero weight
        # (base.py:513-517, python negative index)
        st[min(2, C - 1), 0, P - 1] = -0.2
        du[min(3, C - 1), 0] = 0.3
    sl = rng.uniform(0, 5, (C, P))
    Ttab = 1 if Ttab_is_one else T
    ro, fa = emu.gf_tables_ml(st, du, 0.0, 0.5, 0.5, 0.5, D, S, Ttab, P)
    DS = D * S
    order = emu.gc_order(ro, C, Ttab, P, S, sort)
    assert sorted(order[order != emu.DEAD].tolist()) == list(range(C))
    tabs = emu.gm_tables(ro, fa, [sl] + slx, order, C, Ttab, P, D, S, nvar=nvar, cap=cap)
    data = rng.standard_normal((T, N))
    wsc = rng.uniform(0.5, 2.0, T)
    ntile = (N + 63) // 64
    mem = emu.Memory()
    a = dict(T=T, P=P, N=N, DS=DS, Ttab=Ttab, rows_per_target=P * DS, mode=mode, ntile=ntile, wscalar=wsc, nvar=nvar,
             nv=tabs["nv"], smax=tabs["smax"], cap=tabs["cap"])
    for i, gx in enumerate(Gx):
        a["G%d" % (i + 1)] = mem.alloc(gx.nbytes, gx)
    a["G"] = mem.alloc(G.nbytes, G)
    for name in ("wtab", "ltab", "dtab"):
        a[name] = mem.alloc(tabs[name].nbytes, tabs[name])
    a["order"] = mem.alloc(order.nbytes + 256, order)
    a["data"] = mem.alloc(data.nbytes, data)
    a["out"] = mem.alloc(C * T * N * 8)
    a["partial"] = mem.alloc(C * T * ntile * 8)
    if mode == 3:
        # rows (W[i,i], W[i,i+1]) of a bidiagonal whitening operator per target, 0 behind the end of the trace
        band = rng.uniform(0.5, 2.0, (T, N, 2))
        band[:, N - 1, 1] = 0.0
        a["band_w"] = mem.alloc(band.nbytes, band)
        a["edges"] = mem.alloc(C * T * ntile * 16)
    ngroups = order.size // emu.CG
    stats = []
    for g in range(ngroups):
        for t in range(T):
            for tile in range(ntile):
                params = [emu.wave_params(w, g, t, tile, a) for w in range(emu.WAVES)]
                wg = emu.Workgroup(mem, nth, emu.lds_bytes(tabs["cap"]), params).run()
                assert all(w.done and not w.idx_en and w.exec == emu.MASK64 for w in wg.waves)
                gt = g * Ttab + (0 if Ttab == 1 else t)
                assert {w.nbarrier for w in wg.waves} == {int(tabs["nv"][gt]) * nvar + 1}
                stats.append(wg)
    # full rows for the reference: with tables per patch the row ids are those of target 0
    rof = ro if Ttab == T else np.stack([ro[:, 0] + t * P * DS for t in range(T)], 1)
    faf = fa if Ttab == T else np.repeat(fa, T, axis=1)
    ref = _reference(G, rof, faf, sl, T, P, N, Gx, slx)
    out = mem.array(a["out"], np.float64, C * T * N).reshape(C, T, N)
    if mode == 0:
        assert np.array_equal(out, ref)
    elif mode == 2:
        assert np.array_equal(out, data[None] - ref)
    elif mode == 3:
        # the canonical order of quadform.hip (k_quadform_band1, the epilogue of k_gfstack_ws): per tile the samples but
        # its last, ascending, + the trace's very last sample; the first / last residual of every tile for the boundary terms
        part = mem.array(a["partial"], np.float64, C * T * ntile).reshape(C, T, ntile)
        edges = mem.array(a["edges"], np.float64, C * T * ntile * 2).reshape(C, T, ntile, 2)
        resid = data[None] - ref
        for c in range(C):
            for t in range(T):
                for tl in range(ntile):
                    n0, nv = tl * 64, min(64, N - tl * 64)
                    q = 0.0
                    for i in range(n0, n0 + nv - 1):
                        y = band[t, i, 0] * resid[c, t, i] + 0.0
                        y = band[t, i, 1] * resid[c, t, i + 1] + y
                        q = y * y + q
                    if n0 + nv == N:
                        y = band[t, N - 1, 0] * resid[c, t, N - 1] + 0.0
                        q = y * y + q
                    assert part[c, t, tl] == q, (c, t, tl, part[c, t, tl], q)
                    assert edges[c, t, tl, 0] == resid[c, t, n0]
                    if n0 + nv < N:
                        assert edges[c, t, tl, 1] == resid[c, t, n0 + nv - 1]
    else:
        part = mem.array(a["partial"], np.float64, C * T * ntile).reshape(C, T, ntile)
        exp = np.zeros_like(part)
        for c in range(C):
            for t in range(T):
                for tl in range(ntile):
                    q = 0.0
                    for i in range(tl * 64, min(N, tl * 64 + 64)):
                        tt = wsc[t] * (data[t, i] - ref[c, t, i])
                        q = tt * tt + q
                    exp[c, t, tl] = q
        assert np.array_equal(part, exp)
    return stats, tabs


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_runs_program_one_group(mode):
    """45 chains (one full consumer wavefront, one partly filled, twelve empty), two tiles (64 + 6 samples), tables per
    target, node-0 wrap, exact-grid durations and start times / durations BELOW the first grid node (the wrapped floor
    node carries weight)"""
    _run(T=2, P=4, D=3, S=6, N=70, C=45, Ttab_is_one=False, mode=mode, sort=bool(mode & 1), nth=mode & 1, seed=5 + mode,
         below_grid=True)


def test_runs_program_shares_row_reads():
    stats, tabs = _run(T=2, P=3, D=2, S=5, N=64, C=80, Ttab_is_one=True, mode=0, sort=True, nth=0, seed=11)
    # four FMAs per chain and step into its accumulator; the pads (empty chain slots here) go to the scratch accumulator,
    # and only up to the first checkpoint behind a wavefront's last chain (early exit): 80 chains = two full wavefronts,
    # one with 6 chains (walks to checkpoint 8: 2 pads) and eleven with none (walk to checkpoint 4: 4 pads)
    wg = stats[0]
    assert sum(w.fma_count - w.pad_fma_count for w in wg.waves) == 80 * 3 * 4
    assert sum(w.pad_fma_count for w in wg.waves) == (2 + 11 * 4) * 3 * 4
    stats, tabs = _run(T=1, P=2, D=2, S=4, N=64, C=519, Ttab_is_one=True, mode=1, sort=True, nth=1, seed=3)
    # LDS-DMA moved every distinct row segment of the group once
    assert stats[0].dma_bytes == int(tabs["ucount"][:2].sum()) * 512
    # D = S = 1: one node, every index wraps onto it
    _run(T=1, P=2, D=1, S=1, N=64, C=3, Ttab_is_one=True, mode=0, sort=False, nth=0, seed=4)
    # a step of one patch and a wavefront full of one cell
    _run(T=1, P=1, D=2, S=3, N=64, C=40, Ttab_is_one=True, mode=2, sort=True, nth=0, seed=9)


@pytest.mark.parametrize("nvar", [2, 3])
def test_programs_with_several_slip_variables(nvar):
    """steps cycle through the slip variables' libraries (loader: base of the step's variable + rows of the patch;
    records: the variable's slips on the same cells)"""
    _run(T=2, P=3, D=2, S=5, N=70, C=45, Ttab_is_one=(nvar == 2), mode=nvar % 3, sort=True, nth=0, seed=21 + nvar, nvar=nvar)


@pytest.mark.parametrize("mode,nvar,ttab1", [(0, 1, True), (1, 2, False), (2, 1, False), (3, 2, True)])
def test_row_passes(mode, nvar, ttab1):
    """a library with more rows per patch than a row buffer holds (here: buffers of 12 slots, 6 x 9 = 54 rows per patch):
    a patch is staged in several passes, a chain takes part in the pass that holds its cell, the other positions of
    its wavefront are pads -- bitwise the one-pass result; the number of steps differs from (group, target) to
    (group, target)"""
    stats, tabs = _run(T=2, P=3, D=6, S=9, N=70, C=90, Ttab_is_one=ttab1, mode=mode, sort=True, nth=mode & 1, seed=31 + mode,
                       below_grid=True, nvar=nvar, cap=12)
    assert tabs["npass"].max() >= 3 and tabs["npass"].min() >= 2
    assert int(tabs["nv"].max()) <= tabs["vmax"] and int(tabs["nv"].min()) == int(tabs["npass"].reshape(-1, 3).sum(1).min())
    # every chain of every patch and variable got its four FMAs exactly once; the rest went to the scratch accumulator
    wg = stats[0]
    real = sum(w.fma_count - w.pad_fma_count for w in wg.waves)
    assert real == 90 * 3 * nvar * 4


def test_row_passes_two_groups_and_a_full_buffer():
    """519 chains (a full group and a group of one chain) with buffers of 16 slots; a one-chain group needs one pass"""
    stats, tabs = _run(T=1, P=2, D=4, S=7, N=64, C=519, Ttab_is_one=True, mode=1, sort=True, nth=1, seed=41, cap=16)
    assert int(tabs["npass"][:2].min()) >= 2 and int(tabs["npass"][2:].max()) == 1
    assert stats[0].dma_bytes == int(tabs["ucount"][:2].sum()) * 512


def test_emulator_catches_a_missing_wait():
    """the emulator's loads are asynchronous (poison in the destination until the s_waitcnt that covers them; LDS-DMA rows
    reach LDS at the loader's wait): a consumer program WITHOUT the waits of its new-cell blocks cannot pass"""
    gen.ABL.add("nolgkm")
    try:
        with pytest.raises((AssertionError, IndexError, RuntimeError)):
            _run(T=1, P=3, D=2, S=5, N=64, C=80, Ttab_is_one=True, mode=0, sort=True, nth=0, seed=11)
    finally:
        gen.ABL.discard("nolgkm")
    _run(T=1, P=3, D=2, S=5, N=64, C=80, Ttab_is_one=True, mode=0, sort=True, nth=0, seed=11)


def test_register_budget():
    """the programs stay inside the registers the kernel may use: 128 VGPRs (16 wavefronts per workgroup = 4 per SIMD)
    and user SGPRs below s96 (VCC, FLAT_SCRATCH, XNACK_MASK above); SGPR pairs used as addresses are even-aligned"""
    import re
    assert gen.V_LAST < 128 and gen.NCONS + gen.NLOAD == 16
    for prog in (gen.consumer(), gen.loader(0), gen.loader(1)):
        for ln in prog:
            for m in re.finditer(r"\b[sv]\[(\d+):(\d+)\]", ln):
                assert int(m.group(1)) % 2 == 0, ln   # (gfx950: SGPR address pairs and VGPR tuples are 64-bit aligned)
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", ln):
                hi = int(m.group(2) or m.group(3))
                assert hi <= 95, ln
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", ln):
                hi = int(m.group(2) or m.group(3))
                assert hi <= gen.V_LAST, ln


def test_committed_include_is_the_generators_output(tmp_path, monkeypatch):
    """beat_amd/csrc/gfruns_asm.inc is generated (tools/gen_gfruns_asm.py) and committed: the file the library is
    built from must be what the generator -- i.e. the programs the emulator tests above run -- writes today"""
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    genruns = importlib.import_module("gen_gfruns_asm")
    committed = open(os.path.join(root, "beat_amd", "csrc", "gfruns_asm.inc")).read()
    monkeypatch.delenv("GR_ABLATIONS", raising=False)
    real_join = os.path.join

    def join3(*a):
        p = real_join(*a)
        return str(tmp_path / "gfruns_asm.inc") if p.endswith("gfruns_asm.inc") else p
    monkeypatch.setattr(genruns.os.path, "join", join3)
    genruns.main()
    monkeypatch.undo()
    assert open(str(tmp_path / "gfruns_asm.inc")).read() == committed


def test_chain_order_twin_is_a_permutation_whatever_the_keys():
    """gc_order (numpy twin of k_gc_cut / k_gc_members / k_gc_order): every chain exactly once per group, with caller keys
    (hypocentre), with start-index keys, with NaN / inf keys (proposals outside everything), for full, partial and
    single-chain groups; batches of several groups are bisected along the key of the wider extent; with keys the first key
    ascends from band to band and the second inside a band, and the number of bands follows the group's extents"""
    rng = np.random.default_rng(3)
    T, P, D, S = 1, 6, 3, 9

    def finite(k):
        return np.where(np.abs(k) <= 1.79e308, k, 0.0)

    def check_cut(groups, f0, f1, lo, hi):
        """groups [lo, hi): the first half of the groups lies below the rest in the key of the wider extent"""
        if hi - lo <= 1:
            return
        ids = np.concatenate(groups[lo:hi])
        half = (hi - lo) // 2
        f = f1 if np.ptp(f1[ids]) > np.ptp(f0[ids]) else f0
        left, right = np.concatenate(groups[lo:lo + half]), np.concatenate(groups[lo + half:hi])
        assert f[left].max() <= f[right].min()
        check_cut(groups, f0, f1, lo, lo + half)
        check_cut(groups, f0, f1, lo + half, hi)

    for C in (1, 36, 37, 518, 519, 1100, 2100, 2600):
        st = rng.uniform(0.0, 0.5 * (S - 1) - 0.01, (C, T, P))
        du = rng.uniform(0.5, 0.5 + 0.5 * (D - 1), (C, P))
        ro, _ = emu.gf_tables_ml(st, du, 0.0, 0.5, 0.5, 0.5, D, S, T, P)
        k0, k1 = rng.uniform(0, 20, C), rng.uniform(0, 10, C)
        for keys in (None, (k0, k1)):
            if keys is not None and C > 5:
                keys[0][3], keys[1][4] = np.nan, np.inf
            order = emu.gc_order(ro, C, T, P, S, True, keys=keys)
            assert order.size == ((C + emu.CG - 1) // emu.CG) * emu.CG
            assert sorted(order[order != emu.DEAD].tolist()) == list(range(C))
            ngr = order.size // emu.CG
            groups = []
            for g in range(ngr):
                ids = order[g * emu.CG:(g + 1) * emu.CG]
                live = ids[ids != emu.DEAD]
                assert live.size == min(emu.CG, C - g * emu.CG)         # full groups first
                assert np.all(ids[:live.size] != emu.DEAD)              # dead slots behind the live ones
                groups.append(live.astype(np.int64))
            if ngr > 1:
                if keys is None:
                    f0, f1 = (ro[:, 0, 0, 3] % S).astype(float), (ro[:, 0, P // 2, 3] % S).astype(float)
                else:
                    f0, f1 = finite(keys[0]), finite(keys[1])
                check_cut(groups, f0, f1, 0, ngr)
                plain = emu.gc_order(ro, C, T, P, S, True, keys=keys, global_members=False)
                assert all(np.all(plain[g * emu.CG:(g + 1) * emu.CG][:min(emu.CG, C - g * emu.CG)] // emu.CG == g) for g in range(ngr))
        if C >= 518:
            order = emu.gc_order(ro, C, T, P, S, True, keys=(k0, k1))
            nbs = set()
            for g in range(C // emu.CG):
                ids = order[g * emu.CG:(g + 1) * emu.CG]
                nw = emu.CG // emu.NCH
                f0, f1 = finite(k0[ids]), finite(k1[ids])
                nb = max(1, min(nw, int(np.rint(np.sqrt(nw * np.ptp(f0) / np.ptp(f1))))))
                nbs.add(nb)
                band = (np.arange(emu.CG) // emu.NCH) * nb // nw
                for b in range(nb - 1):
                    assert f0[band == b].max() <= f0[band == b + 1].min()
                for b in range(nb):
                    assert np.all(np.diff(f1[band == b]) >= 0)
            if C == 518:
                assert nbs == {5}      # one group over 20 x 10: sqrt(14 * 2) = 5.3
            if C == 2100:
                assert len(nbs) > 1 or nbs != {5}, nbs     # (pieces of the fault have other aspect ratios)
    # the members of a cut (device layout: group g at g * cg)
    F0, F1 = rng.uniform(0, 20, 1300), rng.uniform(0, 20, 1300)
    m = emu.gc_cut(F0, F1, 1300, 512, 3)
    assert sorted(m.tolist()) == list(range(1300))
    g0, g1, g2 = m[:512], m[512:1024], m[1024:]
    rest = np.concatenate([g1, g2])
    wide = F1 if np.ptp(F1) > np.ptp(F0) else F0
    assert wide[g0].max() <= wide[rest].min() and g2.size == 276
    assert np.all(np.diff(g0) > 0) and np.all(np.diff(g1) > 0)          # inside a group: by chain id


def test_pass_partition_invariants():
    """the row passes of a (group, target, patch) -- numpy twin of k_gm_tables' count phase, the statement-for-statement
    mirror of the device code -- for random populations on small and large (duration x start-time) grids and buffer
    sizes: every cell has exactly one pass, the passes follow the cell order, a pass's distinct row slots fit the buffer
    and its row requests the two loaders' request lines"""
    rng = np.random.default_rng(8)
    for trial in range(300):
        D, S = int(rng.integers(1, 20)), int(rng.integers(1, 70))
        cap = int(rng.choice([8, 12, 24, 40, 104]))
        S1 = S + 1
        n = int(rng.integers(1, emu.CG + 1))
        spread_d = int(rng.integers(1, D + 1))
        dc = rng.integers(0, spread_d, n) + int(rng.integers(0, D - spread_d + 1))
        sc = rng.integers(0, min(S, int(rng.integers(1, S + 1))), n)
        df = (dc + D - 1) % D
        keys = sorted(set(int(((d * S1 + s_) << 16) | (f * S1 + s_)) for d, s_, f in zip(dc, sc, df)))
        pass_of = emu.passes_along_the_duration_axis(keys, D, S, cap)
        assert set(pass_of) == set(keys)
        order = [pass_of[k] for k in keys]
        assert order == sorted(order) and order[0] == 0 and set(order) == set(range(max(order) + 1))
        for k in range(max(order) + 1):
            slots = set()
            for key in keys:
                if pass_of[key] == k:
                    sb, sa = key >> 16, key & 0xFFFF
                    slots.update((sa, sa + 1, sb, sb + 1))
            assert len(slots) <= max(cap, 4), (D, S, cap, k, len(slots))
            dense = sorted(slots)

            def row_of(sl):
                d, s1 = divmod(sl, S1)
                return d * S + (s1 - 1 if s1 else S - 1)
            nreq, i = 0, 0
            while i < len(dense):
                if i + 1 < len(dense) and 0 < row_of(dense[i + 1]) - row_of(dense[i]) < 256:
                    i += 2
                else:
                    i += 1
                nreq += 1
            assert nreq <= emu.NLOAD * emu.LREQ, (D, S, cap, nreq)


@pytest.mark.parametrize("seed", range(8))
def test_runs_program_random_shapes(seed):
    """seeded random shapes through the emulator (asynchronous loads) against the reference arithmetic: grids from one node
    to 6 x 10, one to 120 chains (partial wavefronts, two groups never -- those have their own tests), one or two slip
    variables, every epilogue, small row buffers (row passes), sorted and unsorted chain order, times below the grid"""
    rng = np.random.default_rng(1000 + seed)
    D, S = int(rng.integers(1, 7)), int(rng.integers(1, 11))
    cap = [None, 12, 16][int(rng.integers(0, 3))]
    if cap is not None and (D * (S + 1) <= cap or S + 1 > cap // 2):
        cap = None       # (a buffer holds at least two start-time nodes of a duration line pair)
    _run(T=int(rng.integers(1, 3)), P=int(rng.integers(1, 5)), D=D, S=S, N=[64, 70][int(rng.integers(0, 2))],
         C=int(rng.integers(1, 121)), Ttab_is_one=bool(rng.integers(0, 2)), mode=int(rng.integers(0, 4)),
         sort=bool(rng.integers(0, 2)), nth=int(rng.integers(0, 2)), seed=seed, below_grid=bool(rng.integers(0, 2)) and D > 1 and S > 1,
         nvar=int(rng.integers(1, 3)), cap=cap)
