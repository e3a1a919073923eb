"""CPU check of the k_gfstack_cell wavefront program (beat_amd/csrc/gfcell_asm.inc).

tools/gfcell_emu.py interprets the instruction list that tools/gen_gfcell_asm.py emits -- all 16
wavefronts of a workgroup with their barriers, the ring of LDS row buffers, the command stream --
and a numpy twin of the table builder; the result is compared with a direct multilinear stack
(reference beat/ffi/base.py:663-704).  Timing / hazards are not modelled; the -m gpu tests run the
real kernel against k_gfstack and the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_gfcell_asm as gen  # noqa: E402
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


def _run(T, P, D, S, N, C, Ttab_is_one, mode, sort, nth, seed, static_acc=False, below_grid=False, nvar=1):
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
    if static_acc:
        tabs = emu.gm_tables(ro, fa, [sl] + slx, order, C, Ttab, P, D, S, runs=(static_acc == "runs"), nvar=nvar)
        wtab, ltab, ucount = tabs[:3]
        dtab = tabs[3] if static_acc == "runs" else None
    else:
        wtab, ltab, ucount = emu.gc_tables(ro, fa, [sl], order, C, Ttab, P, DS)
    data = rng.standard_normal((T, N))
    wsc = rng.uniform(0.5, 2.0, T)
    ntile = (N + 63) // 64
    mem = emu.Memory()
    a = dict(T=T, P=P, N=N, DS=DS, Ttab=Ttab, rows_per_target=P * DS, nsteps=P * nvar, mode=mode, ntile=ntile,
             wscalar=wsc, nvar=nvar)
    for i, gx in enumerate(Gx):
        a["G%d" % (i + 1)] = mem.alloc(gx.nbytes, gx)
    if static_acc:
        a["wstride"], a["ucap"] = (emu.genruns if static_acc == "runs" else emu.genml).WSTRIDE, D * (S + 1)
    a["G"] = mem.alloc(G.nbytes, G)
    a["wtab"] = mem.alloc(wtab.nbytes, wtab)
    a["ltab"] = mem.alloc(ltab.nbytes, ltab)
    a["order"] = mem.alloc(order.nbytes + 256, order)
    if static_acc == "runs":
        a["dtab"] = mem.alloc(dtab.nbytes, dtab)
    a["data"] = mem.alloc(data.nbytes, data)
    a["out"] = mem.alloc(C * T * N * 8)
    a["partial"] = mem.alloc(C * T * ntile * 8)
    ngroups = order.size // emu.CG
    stats = []
    for g in range(ngroups):
        for t in range(T):
            for tile in range(ntile):
                params = [emu.wave_params(w, g, t, tile, a) for w in range(emu.WAVES)]
                nlds = emu.lds_bytes_ml(D * (S + 1)) if static_acc else emu.lds_bytes(DS)
                wg = emu.Workgroup(mem, nth, nlds, params, static_acc=static_acc).run()
                assert all(w.done and not w.idx_en and w.exec == emu.MASK64 for w in wg.waves)
                assert {w.nbarrier for w in wg.waves} == {P * nvar + 1}
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
    return stats, ucount


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_program_one_group(mode):
    """45 chains (one full consumer wavefront, one partly filled, twelve empty), two tiles (64 + 6
    samples), tables per target, node-0 wrap and exact-grid durations included"""
    _run(T=2, P=4, D=3, S=6, N=70, C=45, Ttab_is_one=False, mode=mode, sort=True, nth=mode & 1, seed=5 + mode)


def test_program_tables_per_patch_and_order():
    """tables built once per (chain, patch); results do not depend on the chain order"""
    for sort in (True, False):
        stats, _ = _run(T=2, P=3, D=2, S=5, N=64, C=80, Ttab_is_one=True, mode=0, sort=sort, nth=0, seed=11)
    # every chain of a step costs exactly four indexed FMAs, whatever the batching
    assert sum(w.fma_count for w in stats[0].waves) == 80 * 3 * 4


def test_program_two_groups():
    """519 chains: a full group and a group of one chain"""
    stats, ucount = _run(T=1, P=2, D=2, S=4, N=64, C=519, Ttab_is_one=True, mode=1, sort=True, nth=1, seed=3)
    # LDS-DMA moved every distinct row segment of the group once (plus the three-step prologue overlap)
    assert stats[0].dma_bytes == int(ucount[:2].sum()) * 512


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_static_program_one_group(mode):
    """k_gfstack_ml (static accumulators, dense LDS rows with a wrap slot per duration line): same cases as above
    plus start times / durations BELOW the first grid node, where the wrapped floor node carries weight"""
    _run(T=2, P=4, D=3, S=6, N=70, C=45, Ttab_is_one=False, mode=mode, sort=False, nth=mode & 1, seed=5 + mode,
         static_acc=True, below_grid=True)


def test_static_program_tables_per_patch_and_two_groups():
    stats, _ = _run(T=2, P=3, D=2, S=5, N=64, C=80, Ttab_is_one=True, mode=0, sort=False, nth=0, seed=11, static_acc=True)
    # four FMAs per chain SLOT of a step (dead slots run with zero weights): nothing depends on the data
    assert sum(w.fma_count for w in stats[0].waves) == emu.CG * 3 * 4
    stats, ucount = _run(T=1, P=2, D=2, S=4, N=64, C=519, Ttab_is_one=True, mode=1, sort=False, nth=1, seed=3,
                         static_acc=True)
    assert stats[0].dma_bytes == int(ucount[:2].sum()) * 512
    # D = S = 1: one node, every index wraps onto it
    _run(T=1, P=2, D=1, S=1, N=64, C=3, Ttab_is_one=True, mode=0, sort=False, nth=0, seed=4, static_acc=True)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_runs_program_one_group(mode):
    """k_gfstack_runs (cell order per patch, rows read once per run of chains sharing a cell, accumulators through
    the VGPR index register with the offset unpacked on the scalar side): the cases of the static program"""
    _run(T=2, P=4, D=3, S=6, N=70, C=45, Ttab_is_one=False, mode=mode, sort=bool(mode & 1), nth=mode & 1, seed=5 + mode,
         static_acc="runs", below_grid=True)


def test_runs_program_shares_row_reads():
    stats, _ = _run(T=2, P=3, D=2, S=5, N=64, C=80, Ttab_is_one=True, mode=0, sort=True, nth=0, seed=11, static_acc="runs")
    # four FMAs per chain SLOT and step, but far fewer row reads: 80 chains over 2 x 4 cells
    assert sum(w.fma_count for w in stats[0].waves) == emu.CG * 3 * 4
    stats, ucount = _run(T=1, P=2, D=2, S=4, N=64, C=519, Ttab_is_one=True, mode=1, sort=True, nth=1, seed=3,
                         static_acc="runs")
    assert stats[0].dma_bytes == int(ucount[:2].sum()) * 512
    _run(T=1, P=2, D=1, S=1, N=64, C=3, Ttab_is_one=True, mode=0, sort=False, nth=0, seed=4, static_acc="runs")
    # a step of one patch and a wavefront full of one cell
    _run(T=1, P=1, D=2, S=3, N=64, C=40, Ttab_is_one=True, mode=2, sort=True, nth=0, seed=9, static_acc="runs")


@pytest.mark.parametrize("kind,nvar", [("runs", 2), ("runs", 3), (True, 2)])
def test_programs_with_several_slip_variables(kind, nvar):
    """steps cycle through the slip variables' libraries patch by patch (loader: base of the step's variable + rows of
    the patch; records: the variable's slips on the same cells)"""
    _run(T=2, P=3, D=2, S=5, N=70, C=45, Ttab_is_one=(nvar == 2), mode=nvar % 3, sort=True, nth=0, seed=21 + nvar,
         static_acc=kind, nvar=nvar)


def test_register_budget():
    """the programs stay inside the registers the kernel may use: 128 VGPRs (16 wavefronts per
    workgroup = 4 per SIMD) and user SGPRs below s96 (VCC, FLAT_SCRATCH, XNACK_MASK above); SGPR
    pairs used as addresses are even-aligned"""
    import re
    assert gen.V_LAST < 128 and gen.NCONS + gen.NLOAD == 16 and gen.NQMIN >= 3
    assert emu.genml.V_LAST < 128
    for prog in (gen.consumer(), gen.loader(0), gen.loader(1), emu.genml.consumer(), emu.genruns.consumer()):
        for ln in prog:
            for m in re.finditer(r"\b[sv]\[(\d+):(\d+)\]", ln):
                assert int(m.group(1)) % 2 == 0, ln   # (gfx950: SGPR address pairs and VGPR tuples are 64-bit aligned)
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", ln):
                hi = int(m.group(2) or m.group(3))
                assert hi <= 95, ln
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", ln):
                hi = int(m.group(2) or m.group(3))
                assert hi <= max(gen.V_LAST, emu.genml.V_LAST), ln


def test_committed_include_is_the_generators_output(tmp_path, monkeypatch):
    """beat_amd/csrc/gfcell_asm.inc is generated (tools/gen_gfcell_asm.py) and committed: the file the library is
    built from must be what the generator -- i.e. the program the emulator tests above run -- writes today"""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    gen = importlib.import_module("gen_gfcell_asm")
    committed = open(os.path.join(root, "beat_amd", "csrc", "gfcell_asm.inc")).read()
    monkeypatch.delenv("GC_ABLATIONS", raising=False)
    real_join = os.path.join

    def join(*a):
        p = real_join(*a)
        return str(tmp_path / "gfcell_asm.inc") if p.endswith("gfcell_asm.inc") else p
    monkeypatch.setattr(gen.os.path, "join", join)
    gen.main()
    monkeypatch.undo()
    assert open(str(tmp_path / "gfcell_asm.inc")).read() == committed
    genml = importlib.import_module("gen_gfml_asm")
    committed = open(os.path.join(root, "beat_amd", "csrc", "gfml_asm.inc")).read()
    monkeypatch.delenv("GM_ABLATIONS", raising=False)

    def join2(*a):
        p = real_join(*a)
        return str(tmp_path / "gfml_asm.inc") if p.endswith("gfml_asm.inc") else p
    monkeypatch.setattr(genml.os.path, "join", join2)
    genml.main()
    monkeypatch.undo()
    assert open(str(tmp_path / "gfml_asm.inc")).read() == committed
    genruns = importlib.import_module("gen_gfruns_asm")
    committed = open(os.path.join(root, "beat_amd", "csrc", "gfruns_asm.inc")).read()
    monkeypatch.delenv("GR_ABLATIONS", raising=False)

    def join3(*a):
        p = real_join(*a)
        return str(tmp_path / "gfruns_asm.inc") if p.endswith("gfruns_asm.inc") else p
    monkeypatch.setattr(genruns.os.path, "join", join3)
    genruns.main()
    monkeypatch.undo()
    assert open(str(tmp_path / "gfruns_asm.inc")).read() == committed


def test_chain_order_twin_is_a_permutation_whatever_the_keys():
    """gc_order (numpy twin of k_gc_order): every chain exactly once per group, with caller keys (hypocentre), with
    start-index keys, with NaN / inf keys (proposals outside everything), for full, partial and single-chain groups; with
    keys the first key ascends from band to band and the second inside a band"""
    rng = np.random.default_rng(3)
    T, P, D, S = 1, 6, 3, 9
    for C in (1, 36, 37, 518, 519, 1100):
        st = rng.uniform(0.0, 0.5 * (S - 1) - 0.01, (C, T, P))
        du = rng.uniform(0.5, 0.5 + 0.5 * (D - 1), (C, P))
        ro, _ = emu.gf_tables_ml(st, du, 0.0, 0.5, 0.5, 0.5, D, S, T, P)
        k0, k1 = rng.uniform(0, 20, C), rng.uniform(0, 20, C)
        for keys in (None, (k0, k1)):
            if keys is not None and C > 5:
                keys[0][3], keys[1][4] = np.nan, np.inf
            order = emu.gc_order(ro, C, T, P, S, True, keys=keys)
            assert order.size == ((C + emu.CG - 1) // emu.CG) * emu.CG
            assert sorted(order[order != emu.DEAD].tolist()) == list(range(C))
            ngr = order.size // emu.CG
            for g in range(ngr):
                ids = order[g * emu.CG:(g + 1) * emu.CG]
                live = ids[ids != emu.DEAD]
                assert live.size == min(emu.CG, C - g * emu.CG)         # full groups first
                assert np.all(ids[:live.size] != emu.DEAD)              # dead slots behind the live ones
            if ngr > 1:
                # several groups: the batch is cut in the order of the first key (a slice of the fault per group)
                f0 = (ro[:, 0, 0, 3] % S).astype(float) if keys is None else np.where(np.abs(keys[0]) <= 1.79e308, keys[0], 0.0)
                for g in range(ngr - 1):
                    a_, b_ = order[g * emu.CG:(g + 1) * emu.CG], order[(g + 1) * emu.CG:(g + 2) * emu.CG]
                    assert f0[a_[a_ != emu.DEAD]].max() <= f0[b_[b_ != emu.DEAD]].min()
                plain = emu.gc_order(ro, C, T, P, S, True, keys=keys, global_members=False)
                assert all(np.all(plain[g * emu.CG:(g + 1) * emu.CG][:min(emu.CG, C - g * emu.CG)] // emu.CG == g) for g in range(ngr))
        if C >= 518:
            order = emu.gc_order(ro, C, T, P, S, True, keys=(k0, k1))
            ids = order[:emu.CG]
            nw = emu.CG // emu.NCH
            band = (np.arange(emu.CG) // emu.NCH) * 4 // nw
            f0 = np.where(np.abs(k0[ids]) <= 1.79e308, k0[ids], 0.0)
            f1 = np.where(np.abs(k1[ids]) <= 1.79e308, k1[ids], 0.0)
            for b in range(3):
                assert f0[band == b].max() <= f0[band == b + 1].min()
            for b in range(4):
                assert np.all(np.diff(f1[band == b]) >= 0)
