"""Row passes (round 5): libraries on fine (duration x start-time) grids -- the reference's tutorial uses durations
0-4 s every 0.25 s (D = 17, docs/examples/FFI_kinematic.rst:185-195) and start times every 0.5 s over tens of seconds
(beat/config.py:1906-1909) -- where a chain group touches more distinct library rows per patch than an LDS row
buffer holds.  The loader/consumer kernel (nearest neighbour) and the runs kernel (multilinear) then stage a patch
in several passes; results stay bitwise those of the streaming kernel (reference arithmetic beat/ffi/base.py:607-709)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import beat_amd
    return beat_amd.get_context(0)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _lib(ctx, G, st_dt=0.5, du_dt=0.25, du_min=0.0):
    from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    T, P, D, S, N = G.shape
    gf = SeismicGFLibrary(SeismicGFLibraryConfig(dimensions=G.shape, starttime_sampling=st_dt, duration_sampling=du_dt,
                                                 starttime_min=0.0, duration_min=du_min))
    gf.setup(T, P, D, S, N, allocate=True)
    gf._gfmatrix[:] = G
    gf.init_optimization(ctx)
    return gf


@pytest.mark.parametrize("C", [512, 530, 1100])
def test_nn_row_passes_equal_streaming_kernel(ctx, orc, monkeypatch, C):
    """tutorial grid (17 x 41 = 697 rows per patch), chains spread over all of it: ~360 distinct rows per 512-chain
    group and patch = 4 passes of <= 96 rows through k_gfstack_ws; partial sample tile (N = 130), several groups"""
    T, P, D, S, N = 2, 7, 17, 41, 130
    rng = np.random.default_rng(C)
    G = rng.standard_normal((T, P, D, S, N))
    gf = _lib(ctx, G)
    dur = rng.uniform(0.0, 4.0, (C, P))
    st = rng.uniform(0.0, 19.9, (C, T, P))
    sl = rng.uniform(-1, 5, (C, P))
    # one patch with few rows (a single pass between patches of several) and one whose chains all agree
    st[:, :, 2] = rng.uniform(3.0, 4.9, (C, T))
    dur[:, 2] = rng.uniform(1.0, 1.2, C)
    st[:, :, 5] = 7.3
    dur[:, 5] = 2.1
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    a = gf.stack_all_batch(dur, st, sl)
    assert ctx.last_kernel().startswith("k_gfstack<0,"), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    for order in ("1", "0"):
        monkeypatch.setenv("BEATAMD_GS_ORDER", order)
        monkeypatch.setenv("BEATAMD_GS_NTHINT", order)
        monkeypatch.setenv("BEATAMD_WS_MAP", order)    # distinct rows by presence map (default) / by ranking the row ids
        b = gf.stack_all_batch(dur, st, sl)
        assert ctx.last_kernel() == "k_gfstack_ws<1,0,3,%s>" % order, ctx.last_kernel()
        st_ = ctx.gf_group_stats()
        plan = ctx.gf_plan()
        assert st_["max_rows"] > 192 and plan["max_passes"] >= 3 and 1.0 < plan["mean_passes"] < plan["max_passes"], (st_, plan)
        assert "passes" in plan["plan"]
        assert np.array_equal(a, b), (C, order)
    for c in (0, 511, C - 1):
        ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.0, 0.25, 0.0, 0.5)
        assert np.abs(a[c] - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("nvar,cov,shifts", [(1, "scalar", False), (2, "toeplitz", True), (2, "scalar", True)])
def test_fused_model_on_a_fine_grid_nn(ctx, monkeypatch, nvar, cov, shifts):
    """the fused log-likelihood (scalar-weight epilogue / residual store for dense W) through row passes: 600 chains of a
    model whose library has 17 durations x 41 start times; against the streaming kernel (1e-11: two slip variables
    accumulate in another order there only if the kernels differed -- they must not) and, bitwise, a sub-batch"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((5,), (6,), (1.0,), T=3, N=96, D=17, S=41, du_min=0.0, du_dt=0.25, covariance=cov,
                         slip_varnames=("uparr", "uperp")[:nvar], station_shifts=shifts, time_bounds=(0.0, 12.0))
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = 600
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<0,"), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    B = f.batch(Q)
    # (Toeplitz covariance: bidiagonal operator -> the misfit rides in the kernel's epilogue, mode 3; round 6)
    assert ctx.last_kernel().startswith("k_gfstack_ws<1,%d,3," % (1 if cov == "scalar" else 3)), ctx.last_kernel()
    assert ctx.gf_plan()["max_passes"] >= 2, ctx.gf_plan()
    assert np.isfinite(B).all()
    # (the streaming kernel sums the misfit over 512-sample tiles, the chain-shared kernels over 64-sample tiles)
    np.testing.assert_allclose(A, B, rtol=1e-11, atol=1e-9)
    assert np.array_equal(f.batch(Q[:512]), B[:512])
    # several groups are cut along the hypocentre (a slice of the fault per group): scheduling only
    monkeypatch.setenv("BEATAMD_GC_GLOBAL", "0")
    assert np.array_equal(f.batch(Q), B)
    monkeypatch.delenv("BEATAMD_GC_GLOBAL")
    # the lane <-> chain kernel with 128-chain groups (its row buffers hold a group's whole bound) has the same epilogue
    monkeypatch.setenv("BEATAMD_GS_CG", "128")
    B2 = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_dma<2,1,"), ctx.last_kernel()
    assert np.array_equal(B2, B)


def test_float_storage_kernel_with_row_passes(ctx, monkeypatch):
    """k_gfstack_ws32 (float copy of the library, pair gather) walks the same pass tables"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((5,), (6,), (1.0,), T=3, N=100, D=17, S=41, du_min=0.0, du_dt=0.25, time_bounds=(0.0, 12.0))
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 700)
    f.round_libraries_to_f32()
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    B = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws32<1,3,"), ctx.last_kernel()
    assert ctx.gf_plan()["max_passes"] >= 2
    f.set_f32(False)
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws<1,1,3,"), ctx.last_kernel()
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A0 = f.batch(Q)
    assert np.array_equal(A, B)
    np.testing.assert_allclose(A0, A, rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("C", [518, 530, 1100])
def test_ml_row_passes_equal_streaming_kernel(ctx, orc, monkeypatch, C):
    """multilinear on the tutorial grid (17 x 42 = 714 dense slots per patch, a row buffer holds 104): chains over all
    duration lines and (a) a band of start times -- 2-3 passes along the duration axis --, (b) the whole start-time axis
    -- ~15 passes, more than the tables are sized for by default: the streaming kernel stands in --, (c) the same with
    larger tables.  Bitwise the streaming kernel; times on / below the first grid node included"""
    T, P, D, S, N = 2, 7, 17, 41, 130
    rng = np.random.default_rng(100 + C)
    G = rng.standard_normal((T, P, D, S, N))
    gf = _lib(ctx, G)
    sl = rng.uniform(-1, 5, (C, P))
    for case in ("band", "all", "all_big_tables"):
        dur = rng.uniform(0.0, 4.0, (C, P))
        st = rng.uniform(5.0, 9.0, (C, T, P)) if case == "band" else rng.uniform(0.0, 19.9, (C, T, P))
        st[0::11, :, 0] = 0.0          # on node 0: the floor node wraps (weight 0)
        st[1::13, :, 1] = -0.2         # below node 0: the last node enters with weight
        dur[2::7, 3] = 0.0
        st[:, :, 5] = 7.3              # a patch whose chains all agree: one pass between patches of several
        dur[:, 5] = 2.1
        monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
        a = gf.stack_all_batch(dur, st, sl, interpolation="multilinear")
        assert ctx.last_kernel().startswith("k_gfstack<1,"), ctx.last_kernel()
        monkeypatch.delenv("BEATAMD_GF_KERNEL")
        if case == "all_big_tables":
            monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "40")
        for order in ("1", "0"):
            monkeypatch.setenv("BEATAMD_GS_ORDER", order)
            monkeypatch.setenv("BEATAMD_GS_NTHINT", order)
            monkeypatch.setenv("BEATAMD_GC_SORT", order)
            b = gf.stack_all_batch(dur, st, sl, interpolation="multilinear")
            assert ctx.last_kernel() == "k_gfstack_runs<0,%s>" % order, ctx.last_kernel()
            plan = ctx.gf_plan()
            assert plan["max_passes"] >= (2 if case == "band" else 8) and "passes" in plan["plan"], plan
            assert np.array_equal(a, b), (C, case, order)
        monkeypatch.delenv("BEATAMD_GR_PASS_ALLOC", raising=False)
        for c in (0, 1, 2, 517, C - 1):
            ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.0, 0.25, 0.0, 0.5, "multilinear")
            assert np.abs(a[c] - ref).max() <= 1e-11 * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize("nvar,cov,shifts", [(1, "scalar", False), (2, "toeplitz", True), (3, "scalar", True)])
def test_fused_model_on_a_fine_grid_ml(ctx, monkeypatch, nvar, cov, shifts):
    """the fused log-likelihood through k_gfstack_runs with row passes (scalar-weight epilogue / residual store), one to
    three slip variables, station shifts (tables per target: another number of steps per target): 600 chains of a model
    whose library has 17 durations x 41 start times; against the streaming kernel (1e-11), itself with other buffers
    (bitwise) and a sub-batch (bitwise)"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((5,), (6,), (1.0,), T=3, N=96, D=17, S=41, du_min=0.0, du_dt=0.25, covariance=cov,
                         slip_varnames=("uparr", "uperp", "utens")[:nvar], station_shifts=shifts, time_bounds=(0.0, 3.0),
                         interpolation="multilinear")
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = 600
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<1,"), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    mode = 1 if cov == "scalar" else 3     # (the "exponential" Toeplitz structure: bidiagonal operator, misfit in the epilogue)
    B = f.batch(Q)
    plan = ctx.gf_plan()
    assert ctx.last_kernel().startswith("k_gfstack_runs<%d," % mode) and 2 <= plan["max_passes"] <= 6, (ctx.last_kernel(), plan)
    assert np.isfinite(B).all()
    np.testing.assert_allclose(A, B, rtol=1e-11, atol=1e-9)
    assert np.array_equal(f.batch(Q[:518]), B[:518])
    monkeypatch.setenv("BEATAMD_GR_CAP", "40")
    monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "40")
    B2 = f.batch(Q)
    assert ctx.gf_plan()["max_passes"] > plan["max_passes"]
    assert np.array_equal(B2, B)
    # tables sized for ONE pass per patch overflow: the runs kernel returns at once, the streaming kernel stands in (its own
    # tile sums / residuals, the guarded sum of tiles) -- the streaming kernel's bits, no host synchronisation in between
    monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "1")
    B3 = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<%d," % mode)
    assert np.array_equal(B3, A)
    monkeypatch.delenv("BEATAMD_GR_PASS_ALLOC")
    monkeypatch.delenv("BEATAMD_GR_CAP")
    assert np.array_equal(f.batch(Q), B)      # ... and the flag is cleared for the next call


@pytest.mark.parametrize("interp,C", [("nearest_neighbor", 600), ("multilinear", 300), ("nearest_neighbor", 40)])
def test_targets_of_a_station_share_their_index_tables(ctx, monkeypatch, interp, C):
    """station corrections are per STATION, a station has several channels = targets (heart.py:2941-2950 repeats the
    station indices per channel): the index tables (row ids, LDS slots, passes) are built per distinct shift variable,
    not per target -- 7 targets on 3 stations here.  Bitwise the per-target tables (BEATAMD_GF_TINV=0) in every kernel
    family, and the oracle on sampled chains"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((5,), (6,), (1.0,), T=7, N=96, D=3, S=25, station_shifts=True, interpolation=interp,
                         slip_varnames=("uparr", "uperp"), covariance="toeplitz")
    prob, host = build_problem(spec)
    assert len(set(np.asarray(host["time_shifts"][1]).tolist())) == 3
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    B = f.batch(Q)
    name = ctx.last_kernel()
    monkeypatch.setenv("BEATAMD_GF_TINV", "0")
    A = f.batch(Q)
    assert ctx.last_kernel() == name
    monkeypatch.delenv("BEATAMD_GF_TINV")
    assert np.array_equal(A, B), name
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    S_ = f.batch(Q)
    np.testing.assert_allclose(S_, B, rtol=1e-11, atol=1e-9)
    for c in (0, C - 1):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(B[c], ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("C,cpg", [(600, 512), (1300, 512), (2100, 518), (2600, 512), (4096, 518), (4096, 512), (5000, 518), (8192, 512), (3, 2), (130, 128),
                                   (10000, 512), (10000, 518), (8193, 512), (20000, 518), (300, 2)])
def test_chain_groups_of_a_batch_equal_the_numpy_twin(ctx, C, cpg):
    """k_gc_cut / k_gc_members (batches of several chain groups are bisected along the hypocentre key of the wider extent:
    a compact piece of the fault per group) against tools/gfcell_emu.gc_cut, index for index; odd group counts, partial
    last group, ties, NaN / inf keys.  Scheduling only -- the stacking tests pin that results do not depend on it"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gfcell_emu as emu
    rng = np.random.default_rng(C + cpg)
    k0, k1 = rng.uniform(0, 20, C), rng.uniform(0, 20 if C % 2 else 9, C)
    k0[rng.integers(0, C, 40)] = k0[C // 2]      # ties
    k1[rng.integers(0, C, 40)] = k1[C // 3]
    if C > 12:
        k0[5], k1[11], k1[12] = np.nan, np.inf, -np.inf
    got = ctx.gf_chain_groups(k0, k1, cpg)
    ng = (C + cpg - 1) // cpg
    fin = [np.where(np.abs(k) <= 1.79e308, k, 0.0) for k in (k0, k1)]
    want = np.full(ng * cpg, 0xffffffff, dtype=np.uint32)
    want[:C] = emu.gc_cut_chunked(fin[0], fin[1], C, cpg)      # (one chunk up to 8192 chains / 64 groups: = gc_cut)
    assert np.array_equal(got.ravel(), want)
    live = got.ravel()[:C]
    assert sorted(live.tolist()) == list(range(C))


@pytest.mark.parametrize("interp", ["nearest_neighbor", "multilinear"])
def test_batches_of_many_groups_cut_into_pieces_of_the_fault(ctx, monkeypatch, interp):
    """2600 chains = 6 chain groups (three levels of the cut, a 3 + 3 and 1 + 2 split, partial last group) of the fused
    model: bitwise the batch taken group by group as the chains come (BEATAMD_GC_GLOBAL=0) and the streaming kernel's
    likelihoods to rounding"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((6,), (7,), (1.0,), T=2, N=64, D=3, S=25, interpolation=interp, time_bounds=(0.0, 2.0))
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = 2600
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    if interp == "nearest_neighbor":
        monkeypatch.setenv("BEATAMD_GS_CG", "512")   # (a library this small would be stacked in 128-chain groups)
    B = f.batch(Q)
    name = ctx.last_kernel()
    assert name.startswith("k_gfstack_ws<" if interp == "nearest_neighbor" else "k_gfstack_runs<"), name
    monkeypatch.setenv("BEATAMD_GC_GLOBAL", "0")
    assert np.array_equal(f.batch(Q), B)
    monkeypatch.delenv("BEATAMD_GC_GLOBAL")
    assert np.array_equal(f.batch(Q[:1100]), B[:1100])
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<"), ctx.last_kernel()
    np.testing.assert_allclose(A, B, rtol=1e-11, atol=1e-9)
