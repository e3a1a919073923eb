"""GPU parity tests: the HIP path (through the C ABI, via beat_amd.engine) against the CPU
oracle and the committed golden vectors.  Run on the MI355X box: pytest -m gpu.

Tolerances: indices bit-exact; rupture times <= 1e-12 s abs (sqrt vs glibc pow, SURVEY
A.8); synthetics and log-likelihoods <= 1e-6 rel as BASELINE.json's north_star states
(observed ~1e-13)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-6  # north_star tolerance for synthetics / logp


@pytest.fixture(scope="module")
def ctx():
    import beat_amd
    c = beat_amd.get_context(0)
    yield c


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


# ----------------------------------------------------------------------------- sweep
def test_sweep_golden_times_and_indices(ctx, orc):
    g = load_golden("sweep")
    for name in g["names"]:
        slow = g[name + "_slow"]
        psz, hd, hs, nd, ns = g[name + "_meta"]
        nd, ns = int(nd), int(ns)
        out = ctx.fast_sweep_batch(slow.reshape(1, -1), psz, [int(hd)], [int(hs)], nd, ns)[0]
        ref = g[name + "_c"]  # the reference C extension's output
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12, err_msg=name)
        # bit-exact rupture-time INDEXING (north_star): nn and multilinear grid indices
        for dt in (0.5, 0.25, 0.1):
            assert np.array_equal(orc.time2idx(out, 0.0, dt)[0], orc.time2idx(ref, 0.0, dt)[0])
            assert np.array_equal(orc.time2idx(out, 0.0, dt, "multilinear")[0],
                                  orc.time2idx(ref, 0.0, dt, "multilinear")[0])


def test_sweep_tie_fixtures_indices_bit_exact(ctx, orc):
    """slowness fields that put t / dt on exact k + 0.5 ties (dyadic slowness x patch
    size): the device sweep must reproduce those times EXACTLY (any contraction or re-association
    moves them off the tie) and the grid indices of the reference (round-half-even / ceil)"""
    g = load_golden("sweep_ties")
    n_ties = 0
    for name in g["names"]:
        slow = g[name + "_slow"]
        psz, hd, hs, nd, ns = g[name + "_meta"]
        nd, ns = int(nd), int(ns)
        out = ctx.fast_sweep_batch(slow.reshape(1, -1), psz, [int(hd)], [int(hs)], nd, ns)[0]
        ref = g[name + "_c"]
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12, err_msg=name)
        for dt in (0.5, 0.25):
            tie = g[name + "_ties_%g" % dt]
            n_ties += int(tie.sum())
            assert np.array_equal(out[tie], ref[tie]), name          # ties are exact sums
            assert np.array_equal(orc.time2idx(out, 0.0, dt)[0], g[name + "_idx_nn_%g" % dt]), name
            assert np.array_equal(orc.time2idx(out, 0.0, dt, "multilinear")[0], g[name + "_idx_ml_%g" % dt])
    assert n_ties >= 50
    # through the fused model: the index tables the stacking kernel builds from these times select
    # the reference's rows -- one patch-wide library whose rows encode (duration, starttime) index
    from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    name = "tie_two_media"
    slow = g[name + "_slow"]
    psz, hd, hs, nd, ns = g[name + "_meta"]
    P, S = slow.size, 64
    gf = SeismicGFLibrary(SeismicGFLibraryConfig(dimensions=(1, P, 1, S, 2), starttime_sampling=0.5,
                                                 duration_sampling=0.5, starttime_min=0.0, duration_min=0.5))
    gf.setup(1, P, 1, S, 2, allocate=True)
    gf._gfmatrix[0, :, 0, :, 0] = np.arange(S)[None, :]          # sample 0 of a row = its starttime index
    gf._gfmatrix[0, :, 0, :, 1] = 1.0
    gf.init_optimization(ctx)
    t = ctx.fast_sweep_batch(slow.reshape(1, -1), psz, [int(hd)], [int(hs)], int(nd), int(ns))
    for p in range(P):
        sl = np.zeros((1, P))
        sl[0, p] = 1.0
        syn = gf.stack_all_batch(np.full((1, P), 0.5), t.reshape(1, 1, P), sl)
        assert syn[0, 0, 0] == g[name + "_idx_nn_0.5"][p] and syn[0, 0, 1] == 1.0


def test_sweep_batch_random_vs_oracle(ctx, orc):
    rng = np.random.default_rng(5)
    for nd, ns in [(20, 20), (7, 31), (64, 3), (70, 70), (1, 1), (2, 90)]:
        C = 37
        slow = 1.0 / rng.uniform(0.5, 6.0, (C, nd * ns))
        hd = rng.integers(0, nd, C).astype(np.int32)
        hs = rng.integers(0, ns, C).astype(np.int32)
        out = ctx.fast_sweep_batch(slow, 1.5, hd, hs, nd, ns)
        for c in range(C):
            ref = orc.fast_sweep(slow[c], 1.5, hd[c], hs[c], nd, ns)
            np.testing.assert_allclose(out[c], ref, rtol=0, atol=1e-12)
            assert np.array_equal(orc.time2idx(out[c], 0.0, 0.5)[0], orc.time2idx(ref, 0.0, 0.5)[0])


def test_sweep_register_upwind_is_bitwise_the_lds_version(ctx, monkeypatch):
    """round 4: grids of at most 64 rows keep the upwind operands of a diagonal in registers (own previous cell, the
    lane below by DPP) and read the rest one diagonal ahead; BEATAMD_SWEEP_V1=1 runs the first, LDS-only loop.  Same
    operands and operations per cell: the times must be equal bit for bit -- all four sweep directions, one-row /
    one-column grids, 64 rows (the limit), hypocentres in corners and on edges, infinitely slow patches."""
    rng = np.random.default_rng(11)
    for nd, ns in [(20, 20), (1, 1), (1, 17), (17, 1), (2, 2), (64, 5), (5, 64), (64, 64), (33, 48), (65, 4)]:
        C = 40
        slow = 1.0 / rng.uniform(0.3, 8.0, (C, nd * ns))
        if nd * ns > 8:
            slow[3, rng.integers(0, nd * ns, 3)] = np.inf
        hd = rng.integers(0, nd, C).astype(np.int32)
        hs = rng.integers(0, ns, C).astype(np.int32)
        hd[:4] = [0, nd - 1, 0, nd - 1]
        hs[:4] = [0, ns - 1, ns - 1, 0]
        monkeypatch.delenv("BEATAMD_SWEEP_V1", raising=False)
        new = ctx.fast_sweep_batch(slow, 1.3, hd, hs, nd, ns)
        monkeypatch.setenv("BEATAMD_SWEEP_V1", "1")
        old = ctx.fast_sweep_batch(slow, 1.3, hd, hs, nd, ns)
        monkeypatch.delenv("BEATAMD_SWEEP_V1", raising=False)
        assert np.array_equal(new, old, equal_nan=True), (nd, ns)


def test_sweep_bad_hypocentre_raises(ctx):
    with pytest.raises(ValueError):
        ctx.fast_sweep_batch(np.ones((1, 12)), 1.0, [3], [0], 3, 4)


def test_sweep_reference_module_signature(ctx):
    """drop-in for fast_sweep_ext.fast_sweep / get_rupture_times_c / Sweeper.perform"""
    from beat_amd.fast_sweeping import fast_sweep, fast_sweep_ext
    from beat_amd.pytensorf import Sweeper
    g = load_golden("sweep")
    slow = g["kat_slow"]
    ref = g["kat_c"]
    a = fast_sweep_ext.fast_sweep(slow.ravel().copy(), 10.0, 3, 2, 6, 4)
    b = fast_sweep.get_rupture_times_c(slow.ravel().copy(), 10.0, 4, 6, 2, 3)
    out = [[None]]
    Sweeper(10.0, 6, 4, "c").perform(None, (slow.ravel(), 3, 2), out)
    for x in (a, b, out[0][0]):
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-12)
    with pytest.raises(AttributeError):
        fast_sweep_ext.fast_sweep(slow.ravel().astype(np.float32), 10.0, 3, 2, 6, 4)
    with pytest.raises(AttributeError):
        fast_sweep_ext.fast_sweep([1.0, 2.0], 10.0, 0, 0, 1, 2)


# ----------------------------------------------------------------------------- stacking
def _make_lib(ctx, G, st_min, st_dt, du_min, du_dt):
    from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    T, P, D, S, N = G.shape
    gf = SeismicGFLibrary(SeismicGFLibraryConfig(dimensions=G.shape, starttime_sampling=st_dt,
                                                 duration_sampling=du_dt, starttime_min=st_min,
                                                 duration_min=du_min))
    gf.setup(T, P, D, S, N, allocate=True)
    gf._gfmatrix[:] = G
    gf.init_optimization(ctx)
    return gf


@pytest.mark.parametrize("interp,tag", [("nearest_neighbor", "nn"), ("multilinear", "ml")])
def test_stack_all_golden(ctx, interp, tag):
    g = load_golden("stack_all")
    st_min, st_dt, du_min, du_dt = g["cfg"]
    gf = _make_lib(ctx, g["G"], st_min, st_dt, du_min, du_dt)
    T = g["G"].shape[0]
    for k in range(int(g["ncase"])):
        out = gf.stack_all(durations=g["c%d_dur" % k], starttimes=g["c%d_st" % k],
                           slips=g["c%d_sl" % k], targetidxs=np.atleast_2d(np.arange(T)).T,
                           interpolation=interp)
        ref = g["c%d_%s_out" % (k, tag)]
        np.testing.assert_allclose(out, ref, rtol=RTOL, atol=1e-9)
        assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("N", [16, 37, 512, 1000, 1026])
@pytest.mark.parametrize("interp", ["nearest_neighbor", "multilinear"])
def test_stack_all_batch_vs_oracle(ctx, orc, N, interp):
    rng = np.random.default_rng(N)
    T, P, D, S = 3, 21, 3, 6
    G = rng.standard_normal((T, P, D, S, N))
    gf = _make_lib(ctx, G, 0.0, 0.5, 0.5, 0.25)
    C = 5
    dur = rng.uniform(0.5, 1.0, (C, P))
    st = rng.uniform(0.0, 2.5, (C, T, P))
    sl = rng.uniform(0, 5, (C, P))
    out = gf.stack_all_batch(dur, st, sl, interpolation=interp)
    for c in range(C):
        ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.25, 0.0, 0.5, interp)
        np.testing.assert_allclose(out[c], ref, rtol=RTOL, atol=1e-9)
        assert np.abs(out[c] - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("order,cgroup", [("0", "128"), ("1", "2"), ("1", "3"), ("1", "128")])
def test_stack_block_orders_identical(ctx, orc, monkeypatch, order, cgroup):
    """the block -> (chain, target, tile) mapping is a pure scheduling choice"""
    monkeypatch.setenv("BEATAMD_GF_ORDER", order)
    monkeypatch.setenv("BEATAMD_GF_CGROUP", cgroup)
    rng = np.random.default_rng(77)
    T, P, D, S, N = 4, 9, 2, 5, 1100
    G = rng.standard_normal((T, P, D, S, N))
    gf = _make_lib(ctx, G, 0.0, 0.5, 0.5, 0.5)
    C = 7
    dur = rng.uniform(0.5, 1.0, (C, P))
    st = rng.uniform(0.0, 2.0, (C, T, P))
    sl = rng.uniform(0, 5, (C, P))
    out = gf.stack_all_batch(dur, st, sl)
    for c in range(C):
        ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.5, 0.0, 0.5)
        assert np.abs(out[c] - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("C", [5, 64, 70, 130, 300, 530])
@pytest.mark.parametrize("interp", ["nearest_neighbor", "multilinear"])
def test_shared_row_kernel_equals_streaming_kernel(ctx, orc, monkeypatch, C, interp):
    """the chain-shared kernels (distinct rows staged once per chain group: k_gfstack_dma with
    ds_read_b64 / ds_read_b128 layouts, k_gfstack_ws, k_gfstack_runs with and without row passes) vs k_gfstack:
    bitwise equal synthetics for one slip variable, and all equal to the oracle"""
    rng = np.random.default_rng(C)
    T, P, D, S, N = 3, 17, 3, 6, 200
    G = rng.standard_normal((T, P, D, S, N))
    gf = _make_lib(ctx, G, 0.0, 0.5, 0.5, 0.25)
    dur = rng.uniform(0.5, 1.0, (C, P))
    st = rng.uniform(0.0, 2.5, (C, T, P))
    sl = rng.uniform(0, 5, (C, P))
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    a = gf.stack_all_batch(dur, st, sl, interpolation=interp)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
    b = gf.stack_all_batch(dur, st, sl, interpolation=interp)
    assert np.array_equal(a, b)
    for c in (0, C // 2, C - 1):
        ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.25, 0.0, 0.5, interp)
        assert np.abs(b[c] - ref).max() <= 1e-11 * np.abs(ref).max()
    if interp == "multilinear":
        # k_gfstack_runs (gfcell.hip: lane <-> sample, cell order per patch, rows read once per run of chains sharing a
        # cell, accumulators through the VGPR index: the default from 192 chains on), forced below; chain order and
        # non-temporal requests are scheduling only; with row buffers of 8 slots (18 rows per patch here) every patch is
        # staged in several ROW PASSES -- same bits; with tables sized for one pass per patch they overflow and the
        # streaming kernel stands in
        if C >= 192:
            assert ctx.last_kernel().startswith("k_gfstack_runs<0,"), ctx.last_kernel()
        monkeypatch.setenv("BEATAMD_GS_ML", "1")
        for srt, nth, cap, alloc in (("1", "1", None, None), ("0", "0", None, None), ("1", "0", "8", "12"), ("0", "1", "10", "12"),
                                     ("1", "1", "8", "1")):
            monkeypatch.setenv("BEATAMD_GC_SORT", srt)
            monkeypatch.setenv("BEATAMD_GS_NTHINT", nth)
            for name, val in (("BEATAMD_GR_CAP", cap), ("BEATAMD_GR_PASS_ALLOC", alloc)):
                if val is None:
                    monkeypatch.delenv(name, raising=False)
                else:
                    monkeypatch.setenv(name, val)
            assert np.array_equal(gf.stack_all_batch(dur, st, sl, interpolation=interp), a), (C, srt, nth, cap, alloc)
            assert ctx.last_kernel() == "k_gfstack_runs<0,%s>" % nth, ctx.last_kernel()
            plan = ctx.gf_plan()
            assert (plan["max_passes"] >= 2) == (cap is not None), plan
        for name in ("BEATAMD_GS_ML", "BEATAMD_GC_SORT", "BEATAMD_GS_NTHINT", "BEATAMD_GR_CAP", "BEATAMD_GR_PASS_ALLOC"):
            monkeypatch.delenv(name)
    seen = set()
    for cg in ("64", "128", "256", "512", "1024"):
        monkeypatch.setenv("BEATAMD_GS_CG", cg)
        for dma, nt, order in (("2", "64", "0"), ("2", "64", "1"), ("2", "32", "1"), ("1", "64", "0")):
            if cg == "1024" and dma != "2":
                continue   # 1024-chain groups exist for the LDS-DMA kernel only
            monkeypatch.setenv("BEATAMD_GS_DMA", dma)
            monkeypatch.setenv("BEATAMD_GS_NT", nt)
            monkeypatch.setenv("BEATAMD_GS_ORDER", order)
            monkeypatch.setenv("BEATAMD_GS_WS", "0")
            assert np.array_equal(gf.stack_all_batch(dur, st, sl, interpolation=interp), a), (cg, dma, nt)
            seen.add(ctx.last_kernel())
        if cg == "512":   # loader / consumer wavefronts (single-row interpolation only)
            for order in ("0", "1"):
                monkeypatch.setenv("BEATAMD_GS_WS", "1")
                monkeypatch.setenv("BEATAMD_GS_ORDER", order)
                monkeypatch.setenv("BEATAMD_GS_NTHINT", order)   # (non-temporal row requests: off / on)
                monkeypatch.setenv("BEATAMD_GS_DMA", "2")
                monkeypatch.setenv("BEATAMD_GS_NT", "64")
                assert np.array_equal(gf.stack_all_batch(dur, st, sl, interpolation=interp), a), (cg, "ws", order)
                seen.add(ctx.last_kernel())
    nrow, w = (4 if interp == "multilinear" else 1), {"64": 1, "128": 2, "256": 4, "512": 8, "1024": 16}
    for cg in w:   # every variant really ran (names as beatamd_ctx_last_kernel reports them)
        assert "k_gfstack_dma<%d,%d,0,32,1>" % (w[cg], nrow) in seen, seen
        if cg != "1024":
            assert "k_gfstack_dma<%d,%d,0,64,1>" % (w[cg], nrow) in seen, seen
            assert "k_gfstack_dma<%d,%d,0,64,0>" % (w[cg], nrow) in seen, seen
    if nrow == 1 and C > 0:
        assert "k_gfstack_ws<1,0,3,0>" in seen and "k_gfstack_ws<1,0,3,1>" in seen, seen


@pytest.mark.parametrize("C", [200, 530])
def test_ml_kernel_wrapped_floor_nodes(ctx, orc, monkeypatch, C):
    """k_gfstack_runs keeps a chain's four rows at two LDS addresses (rows in (duration line, node) order; the floor node of
    start-time node 0 = a copy of the line's LAST node).  Times exactly on node 0 (floor node wraps with weight 0), times BELOW the first node
    (the reference's python index -1 wraps to the last node WITH weight, base.py:513-517) and one-node axes must come
    out as in the streaming kernel (bitwise) and the oracle"""
    rng = np.random.default_rng(40 + C)
    for (T, P, D, S, N) in ((2, 7, 3, 6, 130), (1, 3, 1, 1, 64), (2, 4, 2, 25, 64)):
        G = rng.standard_normal((T, P, D, S, N))
        gf = _make_lib(ctx, G, 0.0, 0.5, 0.5, 0.5)
        dur = rng.uniform(0.5, 0.5 + 0.5 * (D - 1), (C, P)) if D > 1 else np.full((C, P), 0.5)
        st = rng.uniform(0.0, 0.5 * (S - 1) - 0.01, (C, T, P)) if S > 1 else np.zeros((C, T, P))
        if S > 1:
            st[0::7, :, 0] = 0.0                   # on node 0
            st[1::7, :, P - 1] = -0.2               # below node 0: the last node enters with weight 0.4
            st[2::9, 0, 1] = 0.5 * (S - 1)          # on the last node
        if D > 1:
            dur[3::5, 1] = 0.5                      # on duration node 0
            dur[4::11, 0] = 0.3                     # below it
        sl = rng.uniform(0, 5, (C, P))
        monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
        a = gf.stack_all_batch(dur, st, sl, interpolation="multilinear")
        monkeypatch.delenv("BEATAMD_GF_KERNEL")
        monkeypatch.setenv("BEATAMD_GS_ML", "1")
        for cap in (None, "8"):     # one pass per patch / row passes (buffers of 8 slots)
            if cap:
                monkeypatch.setenv("BEATAMD_GR_CAP", cap)
                monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "40")
            b = gf.stack_all_batch(dur, st, sl, interpolation="multilinear")
            assert ctx.last_kernel().startswith("k_gfstack_runs<0,"), ctx.last_kernel()
            assert np.array_equal(a, b), (T, P, D, S, N, cap)
        monkeypatch.delenv("BEATAMD_GS_ML")
        monkeypatch.delenv("BEATAMD_GR_CAP")
        monkeypatch.delenv("BEATAMD_GR_PASS_ALLOC")
        for c in (0, 1, 2, 3, 4, C - 1):
            ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.5, 0.0, 0.5, "multilinear")
            assert np.abs(b[c] - ref).max() <= 1e-11 * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize("interp", ["nearest_neighbor", "multilinear"])
def test_window_slots_with_more_than_32_distinct_rows(ctx, orc, monkeypatch, interp):
    """k_gfstack_dma places the staged rows in LDS by bank window when a chain group uses more than
    32 distinct rows per patch (slots 32 apart share a two-bank window): 75-row library, 530 and
    1100 chains with widely spread start times; bitwise equal to the streaming kernel and to the
    dense-slot numbering, oracle on sampled chains"""
    T, P, D, S, N = 2, 6, 3, 25, 128
    rng = np.random.default_rng(12)
    G = rng.standard_normal((T, P, D, S, N))
    gf = _make_lib(ctx, G, 0.0, 0.5, 0.5, 0.5)
    for C in (530, 1100):
        dur = rng.uniform(0.5, 1.5, (C, P))
        st = rng.uniform(0.0, 11.9, (C, T, P))
        sl = rng.uniform(0, 5, (C, P))
        monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
        a = gf.stack_all_batch(dur, st, sl, interpolation=interp)
        monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
        for cg in ("256", "512", "1024"):
            monkeypatch.setenv("BEATAMD_GS_CG", cg)
            for win in ("1", "0"):
                monkeypatch.setenv("BEATAMD_GS_WIN", win)
                monkeypatch.setenv("BEATAMD_GS_WS", "0")
                b = gf.stack_all_batch(dur, st, sl, interpolation=interp)
                assert ctx.last_kernel().startswith("k_gfstack_dma<"), ctx.last_kernel()
                assert ctx.gf_group_stats()["max_rows"] > 32
                assert np.array_equal(a, b), (C, cg, win)
            if cg == "512" and interp == "nearest_neighbor":
                for win in ("1", "0"):   # three row buffers of 96 slots, four loader wavefronts
                    monkeypatch.setenv("BEATAMD_GS_WIN", win)
                    monkeypatch.setenv("BEATAMD_GS_WS", "1")
                    b = gf.stack_all_batch(dur, st, sl, interpolation=interp)
                    assert ctx.last_kernel().startswith("k_gfstack_ws<1,0,3,"), ctx.last_kernel()
                    assert np.array_equal(a, b), (C, cg, "ws", win)
        for c in (0, 529, C - 1):
            ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.5, 0.0, 0.5, interp)
            assert np.abs(a[c] - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("cg", ["128", "256"])
def test_many_groups_with_residual_store(ctx, monkeypatch, cg):
    """regression: 8192 chains in 32 / 64 chain groups of k_gfstack_dma with the residual written
    out (dense covariance).  The slot/weight loads of the step after the last were still in flight
    when the epilogue built its store addresses in their registers -> memory faults from ~29 groups
    on (timing dependent).  Compared with the streaming kernel and the 512-chain kernel."""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((6,), (6,), (1.0,), T=6, N=256, D=3, S=25, covariance="toeplitz")
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 8192)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
    monkeypatch.setenv("BEATAMD_GS_CG", cg)
    for _ in range(3):
        B = f.batch(Q)
        assert ctx.last_kernel().startswith("k_gfstack_dma<%d,1,2,64,1>" % (int(cg) // 64)), ctx.last_kernel()
        np.testing.assert_allclose(A, B, rtol=1e-11, atol=1e-9)
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    assert np.array_equal(f.batch(Q), B)


def test_more_than_64_distinct_rows_per_step(ctx, orc, monkeypatch):
    """96-row library, 600 chains spread over all of it: a 512-chain group uses > 64 distinct rows
    per patch, beyond the row ids the loader wavefronts (k_gfstack_ws) / the issue block
    (k_gfstack_dma) prefetch into scalar registers -- the overflow loops; bitwise vs streaming"""
    T, P, D, S, N = 2, 5, 3, 32, 128
    rng = np.random.default_rng(5)
    G = rng.standard_normal((T, P, D, S, N))
    gf = _make_lib(ctx, G, 0.0, 0.5, 0.5, 0.5)
    C = 600
    dur = rng.uniform(0.5, 1.5, (C, P))
    st = rng.uniform(0.0, 15.4, (C, T, P))
    sl = rng.uniform(0, 5, (C, P))
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    a = gf.stack_all_batch(dur, st, sl)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    for ws, name in (("1", "k_gfstack_ws<1,0,3,0>"), ("0", "k_gfstack_dma<8,1,0,64,1>")):
        monkeypatch.setenv("BEATAMD_GS_WS", ws)
        b = gf.stack_all_batch(dur, st, sl)
        assert ctx.last_kernel() == name, ctx.last_kernel()
        assert ctx.gf_group_stats()["max_rows"] > 64
        assert np.array_equal(a, b), ws
    for c in (0, 511, 599):
        ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.5, 0.0, 0.5)
        assert np.abs(a[c] - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("name", ["seis_dense_ml_shifts", "joint_multifault", "all_nn_odd_N",
                                  "seis_scalar_nn"])
def test_fused_model_with_shared_row_kernel(ctx, monkeypatch, name):
    """the fused log-likelihood through both stacking kernels, 100 chains"""
    from beat_amd.synthetic import build_problem, draw_population
    from oracle import problem_oracle
    spec = _specs()[name]
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 100)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
    B = f.batch(Q)
    np.testing.assert_allclose(A, B, rtol=1e-11, atol=1e-9)
    for c in (0, 57, 99):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(B[c], ref, rtol=RTOL)
        np.testing.assert_allclose(B[c], ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", ["seis_dense_ml_shifts", "joint_multifault", "seis_scalar_nn"])
def test_fused_model_512_chain_groups(ctx, monkeypatch, name):
    """the fused log-likelihood with 530 chains: 512-chain workgroups of k_gfstack_dma (two slip
    variables, multilinear, station shifts, dense/scalar misfit epilogues) against the streaming
    kernel and sampled chains against the oracle"""
    from beat_amd.synthetic import build_problem, draw_population
    from oracle import problem_oracle
    spec = _specs()[name]
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 530)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    monkeypatch.setenv("BEATAMD_GS_WS", "0")
    B = f.batch(Q)
    np.testing.assert_allclose(A, B, rtol=1e-11, atol=1e-9)
    monkeypatch.setenv("BEATAMD_GS_WS", "1")   # loader / consumer wavefronts where they apply
    B2 = f.batch(Q)
    if "nn" in name:
        assert ctx.last_kernel().startswith("k_gfstack_ws<1,"), ctx.last_kernel()
    assert np.array_equal(B, B2)
    # without station shifts the index tables are built once per (chain, patch) and shared by the
    # targets; BEATAMD_GF_TINV=0 builds them per target as with shifts: same numbers either way
    monkeypatch.setenv("BEATAMD_GF_TINV", "0")
    for ws, kern in (("1", "1"), ("0", "1"), ("0", "0")):
        monkeypatch.setenv("BEATAMD_GS_WS", ws)
        monkeypatch.setenv("BEATAMD_GF_KERNEL", kern)
        B3 = f.batch(Q)
        assert np.array_equal(B3, B if kern == "1" else A), (ws, kern)
    for c in (0, 511, 512, 529):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(B[c], ref, rtol=RTOL)
        np.testing.assert_allclose(B[c], ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("nvar,cov,shifts", [(2, "scalar", False), (2, "toeplitz", True), (3, "scalar", True)])
def test_multilinear_with_several_slip_variables_through_the_runs_kernel(ctx, monkeypatch, nvar, cov, shifts):
    """BEAT's usual FFI set-up samples uparr AND uperp (static_dist_vars, beat/config.py:83) with the default
    multilinear interpolation: 530 chains run through k_gfstack_runs (steps cycle through the variables' libraries
    patch by patch) -- against itself with row passes (bitwise), the streaming kernel (another summation
    order over the variables: 1e-12) and the oracle on sampled chains"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((5,), (6,), (1.0,), T=3, N=136, D=3, S=25, slip_varnames=("uparr", "uperp", "utens")[:nvar],
                         covariance=cov, station_shifts=shifts, interpolation="multilinear")
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 530)
    mode = 1 if cov == "scalar" else 3     # (the "exponential" Toeplitz structure: bidiagonal operator, misfit in the epilogue)
    for name in ("BEATAMD_GF_KERNEL", "BEATAMD_GS_CG", "BEATAMD_GS_ML"):
        monkeypatch.delenv(name, raising=False)
    B = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<%d," % mode), ctx.last_kernel()
    # row passes (buffers of 12 slots; the library has 3 x 25 rows per patch): the same bits
    monkeypatch.setenv("BEATAMD_GR_CAP", "12")
    monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "40")
    B2 = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<%d," % mode) and ctx.gf_plan()["max_passes"] >= 2, ctx.gf_plan()
    assert np.array_equal(B, B2)
    monkeypatch.delenv("BEATAMD_GR_CAP")
    monkeypatch.delenv("BEATAMD_GR_PASS_ALLOC")
    # the chain order is scheduling only: hypocentre keys of the model path (default), start-time index keys, input order
    for knob, val in (("BEATAMD_GC_KEYS", "0"), ("BEATAMD_GC_SORT", "0")):
        monkeypatch.setenv(knob, val)
        B3 = f.batch(Q)
        assert ctx.last_kernel().startswith("k_gfstack_runs<%d," % mode), ctx.last_kernel()
        assert np.array_equal(B, B3), knob
        monkeypatch.delenv(knob)
    # ... also when the keys are NaN / inf for some chains (proposals outside everything): any place will do
    Qn = Q.copy()
    lay = host["layout"]
    Qn[7, lay.offset("nucleation_strike")] = np.nan
    Qn[500, lay.offset("nucleation_dip")] = np.inf
    Bn = f.batch(Qn)
    keep = np.ones(len(Q), dtype=bool)
    keep[[7, 500]] = False
    assert np.array_equal(Bn[keep], B[keep])
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<1,%d," % nvar), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    np.testing.assert_allclose(B, A, rtol=1e-11)
    for c in (0, 511, 512, 529):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(B[c], ref, rtol=RTOL)
        np.testing.assert_allclose(B[c], ref, rtol=1e-9, atol=1e-9)


def test_stack_closed_form_and_linearity(ctx):
    """reference test/test_ffi.py:22-89 recipe: out[t,n] = t*n*sum(slips); plus linearity"""
    T, P, D, S, N = 30, 40, 11, 31, 10
    G = np.empty((T, P, D, S, N))
    G[:] = (np.arange(N)[None, :] * np.arange(T)[:, None])[:, None, None, None, :]
    gf = _make_lib(ctx, G, 0.0, 0.5, 5.0, 0.5)
    rng = np.random.default_rng(0)
    dur = rng.uniform(5.0, 10.0, (2, P))
    st = rng.uniform(0.0, 15.0, (2, T, P))
    sl = rng.random((2, P))
    for interp in ("nearest_neighbor", "multilinear"):
        out = gf.stack_all_batch(dur, st, sl, interpolation=interp)
        for c in range(2):
            expect = np.arange(T)[:, None] * np.arange(N)[None, :] * sl[c].sum()
            np.testing.assert_allclose(out[c], expect, rtol=1e-12, atol=1e-9)
        out2 = gf.stack_all_batch(dur, st, 2.5 * sl, interpolation=interp)
        np.testing.assert_allclose(out2, 2.5 * out, rtol=1e-13, atol=1e-10)


def test_stack_index_out_of_bounds_raises(ctx):
    G = np.zeros((2, 3, 2, 2, 8))
    gf = _make_lib(ctx, G, 0.0, 0.5, 0.5, 0.5)
    with pytest.raises(IndexError):
        gf.stack_all_batch(np.full((1, 3), 5.0), np.zeros((1, 2, 3)), np.ones((1, 3)))
    # the context stays usable afterwards
    out = gf.stack_all_batch(np.full((1, 3), 0.5), np.zeros((1, 2, 3)), np.ones((1, 3)))
    assert out.shape == (1, 2, 8)
    with pytest.raises(NotImplementedError):
        gf.stack_all_batch(np.full((1, 3), 0.5), np.zeros((1, 2, 3)), np.ones((1, 3)),
                           interpolation="cubic")
    with pytest.raises(ValueError):
        gf.stack_all(np.full(3, 0.5), np.zeros((2, 3)), np.ones(3))  # targetidxs missing


def test_geo_stack_golden(ctx):
    from beat_amd.ffi import GeodeticGFLibrary, GeodeticGFLibraryConfig
    g = load_golden("geo_stack")
    G = g["G"]
    gg = GeodeticGFLibrary(GeodeticGFLibraryConfig(dimensions=G.shape))
    gg.setup(*G.shape, allocate=True)
    gg._gfmatrix[:] = G
    np.testing.assert_allclose(gg.stack_all(g["slips"]), g["out"], rtol=1e-12, atol=1e-12)


# ----------------------------------------------------------------------------- likelihood
@pytest.mark.parametrize("M", [10, 48, 205, 214, 400, 1000])
def test_mvn_chol_dense_vs_oracle(ctx, orc, M):
    rng = np.random.default_rng(M)
    nd, C = 3, 70
    Ws, slogs = [], []
    for d in range(nd):
        Cd = (0.3 + d) * orc.exponential_data_covariance(M, 0.5, 2.0) + 0.01 * np.eye(M)
        Ws.append(orc.cov_chol_inverse(Cd))
        slogs.append(orc.cov_log_pdet(Cd))
    Ws[1] = Ws[1] + 1e-3 * rng.standard_normal((M, M))  # a non-triangular weight matrix
    wid = ctx.weights_create_dense(np.stack(Ws), slogs)
    res = rng.standard_normal((C, nd, M))
    hp = rng.uniform(-1, 1, (C, nd))
    out = ctx.mvn_chol_logp_batch(wid, res, hp)
    for c in range(0, C, 7):
        ref = orc.multivariate_normal_chol(Ws, slogs, hp[c], res[c])
        np.testing.assert_allclose(out[c], ref, rtol=RTOL)
        np.testing.assert_allclose(out[c], ref, rtol=1e-11)
    ctx.weights_destroy(wid)


def test_mvn_chol_scalar_and_scipy(ctx, orc):
    """reference test/test_models.py:149-222 toy: C = 0.001 I, hyper 0 -> scipy logpdf"""
    import scipy.stats
    rng = np.random.default_rng(1)
    M, nd, C = 10, 2, 4
    res = 0.03 * rng.standard_normal((C, nd, M))
    Cov = 0.001 * np.eye(M)
    slog = [orc.cov_log_pdet(Cov)] * nd
    wid = ctx.weights_create_scalar([1.0 / np.sqrt(0.001)] * nd, slog, M)
    out = ctx.mvn_chol_logp_batch(wid, res, np.zeros((C, nd)))
    wid2 = ctx.weights_create_dense(np.stack([orc.cov_chol_inverse(Cov)] * nd), slog)
    out2 = ctx.mvn_chol_logp_batch(wid2, res, np.zeros((C, nd)))
    for c in range(C):
        for d in range(nd):
            ref = scipy.stats.multivariate_normal.logpdf(res[c, d], mean=np.zeros(M), cov=Cov)
            np.testing.assert_allclose(out[c, d], ref, rtol=0, atol=1e-6)
            np.testing.assert_allclose(out2[c, d], ref, rtol=0, atol=1e-6)


def test_laplacian_vs_oracle(ctx, orc):
    g = load_golden("laplacian")
    L, logdet = g["20x20_L"], float(g["20x20_logdet"])
    lid = ctx.laplacian_create(L, logdet)
    rng = np.random.default_rng(2)
    C, nvar = 33, 2
    s = rng.uniform(0, 5, (C, nvar, 400))
    h = rng.uniform(-1, 1, C)
    out = ctx.laplacian_logp_batch(lid, s, h)
    for c in range(C):
        ref = sum(orc.laplacian_logp(L, s[c, v], logdet, h[c]) for v in range(nvar))
        np.testing.assert_allclose(out[c], ref, rtol=1e-11)


def test_laquila_geodetic_fixture(ctx, orc):
    """the only real-data fixture: 2 SAR scenes with full covariances"""
    g = load_golden("laquila_geodetic")
    rng = np.random.default_rng(3)
    C = 16
    for i in range(int(g["n"])):
        Cd = g["d%d_C" % i]
        W = orc.cov_chol_inverse(Cd)
        sl = orc.cov_log_pdet(Cd)
        np.testing.assert_allclose(sl, g["d%d_logpdet" % i], rtol=1e-12)
        n = Cd.shape[0]
        wid = ctx.weights_create_dense(W, [sl])
        res = (g["d%d_displacement" % i][None, :] + 0.01 * rng.standard_normal((C, n))) * g["d%d_odw" % i]
        hp = rng.uniform(-0.5, 0.5, (C, 1))
        out = ctx.mvn_chol_logp_batch(wid, res.reshape(C, 1, n), hp)
        for c in range(C):
            ref = orc.mvn_chol_logp(W, res[c], sl, hp[c, 0])
            np.testing.assert_allclose(out[c, 0], ref, rtol=1e-10)


# ----------------------------------------------------------------------------- fused model
def _specs():
    from beat_amd.synthetic import SyntheticSpec
    return {
        "seis_scalar_nn": SyntheticSpec((6,), (5,), (1.0,), T=4, N=64, D=3, S=25),
        "seis_dense_ml_shifts": SyntheticSpec((6,), (5,), (1.0,), T=5, N=48, D=3, S=30,
                                              covariance="toeplitz", station_shifts=True,
                                              interpolation="multilinear", hp_specific=True),
        "joint_multifault": SyntheticSpec((4, 3), (5, 6), (2.0, 2.0), T=3, N=100, D=3, S=40,
                                          slip_varnames=("uparr", "uperp"), covariance="toeplitz",
                                          station_shifts=True, geodetic_nobs=(21, 17)),
        "geo_lap_only": SyntheticSpec((5,), (7,), (1.0,), T=0, N=0, slip_varnames=("uparr", "uperp"),
                                      geodetic_nobs=(30, 12), laplacian=True),
        "all_nn_odd_N": SyntheticSpec((5,), (4,), (1.0,), T=3, N=33, D=3, S=25,
                                      geodetic_nobs=(9,), laplacian=True, hp_specific=True),
    }


@pytest.mark.parametrize("name", list(_specs().keys()))
def test_ffi_logp_batch_vs_oracle(ctx, name):
    from beat_amd.synthetic import build_problem, draw_population
    from oracle import problem_oracle
    spec = _specs()[name]
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = 9
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    LL = f.batch(Q)
    assert LL.shape == (C, f.nllk) and len(prob.out_names) == f.nllk
    for c in range(C):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(LL[c], ref, rtol=RTOL)
        np.testing.assert_allclose(LL[c], ref, rtol=1e-9, atol=1e-9)
    # one-chain call signature of logp_forw_func: list of arrays, like last
    one = f(Q[0])
    assert float(one[-1]) == LL[0, -1]
    assert f.trust_input is True


def test_astep_batch_vs_oracle(ctx):
    from beat_amd.synthetic import build_problem, draw_population
    from oracle import problem_oracle
    spec = _specs()["all_nn_odd_N"]
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    lay = host["layout"]
    lo, up = lay.bounds(host["lower"], host["upper"])
    C = 24
    rng = np.random.default_rng(9)
    Q0 = draw_population(spec, lay, host["lower"], host["upper"], C)
    L0 = f.batch(Q0)
    delta = rng.standard_normal((C, lay.size)) * (up - lo) * 0.02
    delta[3] *= 100.0  # certainly outside the prior box -> rejected without evaluation
    scaling = rng.uniform(0.5, 1.5, C)
    log_u = np.log(rng.random(C))
    beta = 0.3
    Qn, Ln = Q0.copy(), L0.copy()
    acc = f.astep_batch(Qn, Ln, delta, scaling, lo, up, log_u, beta)
    n_acc = 0
    for c in range(C):
        q_ref, l_ref, a_ref = problem_oracle.astep(host, Q0[c], L0[c], delta[c], scaling[c], lo, up,
                                                   log_u[c], beta)
        assert bool(acc[c]) == a_ref
        np.testing.assert_array_equal(Qn[c], q_ref)
        np.testing.assert_allclose(Ln[c], l_ref, rtol=1e-9, atol=1e-9)
        n_acc += a_ref
    assert acc[3] == 0 and 0 < n_acc < C


def test_device_pointer_path_matches_host_path(ctx):
    """torch CUDA tensors in -> torch CUDA tensors out, no host staging"""
    import torch
    from beat_amd.synthetic import build_problem, draw_population
    spec = _specs()["seis_scalar_nn"]
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 6)
    LLh = f.batch(Q)
    Qd = torch.from_numpy(Q).cuda()
    LLd = f.batch(Qd)
    ctx.synchronize()
    assert LLd.is_cuda
    np.testing.assert_array_equal(LLd.cpu().numpy(), LLh)


# ----------------------------------------------------------------------------- samplers on device
def test_smc_on_device_small_ffi(ctx):
    """end to end: SMC stages with the fused device astep; likelihood bookkeeping stays exact"""
    import torch
    from beat_amd.sampler import SMC, smc_sample
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((4,), (4,), (1.0,), T=3, N=32, D=3, S=25, sigma=0.5)
    prob, host = build_problem(spec)
    lay = host["layout"]
    truth = draw_population(spec, lay, host["lower"], host["upper"], 1, seed_offset=77)[0]
    _, ex = problem_oracle.forward(host, truth)
    rng = np.random.default_rng(1)
    prob.wavemaps[0].data[:] = ex["synthetics"] + 0.5 * rng.standard_normal(ex["synthetics"].shape)
    host["data"] = prob.wavemaps[0].data
    f = prob.compile(ctx)
    lo, up = lay.bounds(host["lower"], host["upper"])
    dev = torch.device("cuda", 0)
    step = SMC(f, lo, up, n_chains=96, tune_interval=10, device=dev, random_seed=11)
    pop, lp, betas = smc_sample(25, step)
    assert betas[-1] == 1.0 and len(betas) >= 3
    # the carried likelihood vectors are exactly the forward model at the end points
    LL = f.batch(np.ascontiguousarray(pop))
    np.testing.assert_allclose(LL, lp, rtol=1e-12)
    ref, _ = problem_oracle.forward(host, pop[5])
    np.testing.assert_allclose(lp[5], ref, rtol=1e-9)
    # sampling moved the population towards the data
    prior = draw_population(spec, lay, host["lower"], host["upper"], 96)
    assert lp[:, -1].mean() > f.batch(prior)[:, -1].mean()
    assert 0.0 < np.mean(step.stage_acceptance) < 1.0


def test_pt_on_device_small_ffi(ctx):
    import torch
    from beat_amd.sampler import pt_sample
    from beat_amd.synthetic import SyntheticSpec, build_problem
    spec = SyntheticSpec((4,), (4,), (1.0,), T=2, N=16, D=3, S=25, geodetic_nobs=(6,), laplacian=True)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    lo, up = host["layout"].bounds(host["lower"], host["upper"])
    s, ls, man = pt_sample(f, lo, up, n_chains_posterior=2, n_chains_tempered=4, n_replicas=8,
                           n_samples=64, swap_interval=(5, 10), beta_tune_interval=2,
                           device=torch.device("cuda", 0), random_seed=2)
    assert s.shape == (64, host["layout"].size) and np.isfinite(ls).all()
    # (a batch of 8 takes the streaming kernel, the replicas ran through the chain-shared one:
    # same values up to the summation order of the misfit)
    np.testing.assert_allclose(f.batch(np.ascontiguousarray(s[:8])), ls[:8], rtol=1e-12)
    assert man.sample_count.sum() >= 0 and len(man.history) >= 1


def test_prewhitened_library_matches_dense_weights(ctx):
    """W.G / W.d computed once; per step only the scalar misfit remains"""
    from beat_amd.synthetic import build_problem, draw_population
    from oracle import problem_oracle
    for name in ("seis_dense_ml_shifts", "joint_multifault"):
        spec = _specs()[name]
        prob, host = build_problem(spec)
        f0 = prob.compile(ctx)
        Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 60)
        A = f0.batch(Q)
        prob2, _ = build_problem(spec)
        f1 = prob2.compile(ctx, prewhiten=True)
        assert np.ndim(prob2.wavemaps[0].weights) == 1  # dense W is gone from the per-step path
        B = f1.batch(Q)
        np.testing.assert_allclose(B, A, rtol=1e-9, atol=1e-7)
        ref, _ = problem_oracle.forward(host, Q[7])
        np.testing.assert_allclose(B[7], ref, rtol=RTOL)


def test_astep_bookkeeping_on_torch_stream(ctx):
    """Kernels run on torch's CURRENT stream (the null stream by default), so torch ops that
    produce the inputs / consume the outputs are ordered with them without host syncs.
    Regression: a NULL stream handle once selected the context's private stream -> races."""
    import torch
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((4,), (4,), (1.0,), T=2, N=16, D=3, S=25, geodetic_nobs=(6,), laplacian=True)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    lay = host["layout"]
    lo, up = lay.bounds(host["lower"], host["upper"])
    dev = torch.device("cuda", 0)
    C = 48
    Q0 = torch.from_numpy(draw_population(spec, lay, host["lower"], host["upper"], C)).to(dev)
    L0 = f.batch(Q0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    lo_d, up_d = torch.from_numpy(lo).to(dev), torch.from_numpy(up).to(dev)
    span = torch.from_numpy((up - lo) * 0.02).to(dev)
    acc = torch.zeros(C, dtype=torch.int32, device=dev)
    betas = torch.linspace(1.0, 0.3, C, dtype=torch.float64, device=dev)
    ones = torch.ones(C, dtype=torch.float64, device=dev)
    n_acc = 0
    for i in range(40):
        delta = torch.randn((C, lay.size), generator=gen, device=dev, dtype=torch.float64) * span
        logu = torch.log(torch.rand((C,), generator=gen, device=dev, dtype=torch.float64))
        f.astep_batch(Q0, L0, delta, ones, lo_d, up_d, logu, betas, acc)
        n_acc += int(acc.sum().item())
        Lr = f.batch(Q0)  # the carried likelihood vectors are the forward model at the states
        assert torch.equal(Lr, L0), "step %d" % i
    assert n_acc > 0
    # a side stream works the same way
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        Q1 = Q0 * 1.0
        L1 = f.batch(Q1)
        d = (L1 - L0).abs().max()
    s.synchronize()
    assert float(d) == 0.0


# ----------------------------------------------------------------------------- per-stage setup rows
def test_noise_covariance_estimators_vs_reference_golden(ctx):
    """covariance.py:716-771 on the GPU, bitwise against arrays captured from the reference"""
    from beat_amd import covariance as cov
    g = load_golden("noise_covariance")
    for k in range(int(g["ncase"])):
        d, w = g["c%d_data" % k], int(g["c%d_win" % k])
        np.testing.assert_array_equal(cov.running_window_rms(d, w, "same"), g["c%d_rms_same" % k])
        np.testing.assert_array_equal(cov.running_window_rms(d, w), g["c%d_rms_valid" % k])
        assert np.array_equal(cov.autocovariance(d), g["c%d_autocov" % k])
        assert np.array_equal(cov.non_toeplitz_covariance(d, w), g["c%d_ntc" % k])
    rng = np.random.default_rng(0)
    data = rng.standard_normal((5, 700))
    out = cov.non_toeplitz_covariance_batch(data, 30)
    from oracle import oracle as orc
    for i in range(5):
        np.testing.assert_allclose(out[i], orc.non_toeplitz_covariance(data[i], 30), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("n", [1, 10, 64, 65, 130, 200, 520])
def test_chol_inverse_batch_kernel(ctx, n):
    """beatamd_chol_inverse_batch (blocked factorisation of the exchange-flipped matrix, FP64 MFMA
    GEMMs) against numpy's route of the reference, heart.py:216-253: W = cholesky(inv(C)).T and
    log_pdet; exponential (Toeplitz) and random covariances, sizes around the 64-blocking"""
    rng = np.random.default_rng(n)
    t = np.arange(n)
    Cs = [np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0) * 0.7 + 1e-3 * np.eye(n)]
    A = rng.standard_normal((n, n + 3))
    Cs.append(A @ A.T / (n + 3) + 0.1 * np.eye(n))
    Cs.append(np.diag(rng.uniform(0.5, 2.0, n)))
    Cs = np.stack(Cs)
    W, lp = ctx.chol_inverse_batch(Cs)
    for i in range(3):
        ref = np.linalg.cholesky(np.linalg.inv(Cs[i])).T
        assert np.array_equal(np.tril(W[i], -1), np.zeros((n, n)))          # upper triangular
        np.testing.assert_allclose(W[i], ref, rtol=0, atol=1e-9 * np.abs(ref).max())
        np.testing.assert_allclose(W[i].T @ W[i] @ Cs[i], np.eye(n), rtol=0, atol=1e-8)
        np.testing.assert_allclose(lp[i], np.linalg.slogdet(Cs[i])[1], rtol=1e-11, atol=1e-10)
    # device tensors in -> device tensors out
    import torch
    Wd, lpd = ctx.chol_inverse_batch(torch.from_numpy(Cs).to("cuda:0"))
    assert Wd.is_cuda and np.array_equal(Wd.cpu().numpy(), W) and np.array_equal(lpd.cpu().numpy(), lp)
    # not positive definite: numpy raises LinAlgError, so does the device path; the context stays usable
    bad = Cs.copy()
    bad[1, n // 2, n // 2] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        ctx.chol_inverse_batch(bad)
    W2, _ = ctx.chol_inverse_batch(Cs)
    assert np.array_equal(W2, W)


def test_covariance_class_and_mvn_adaptor(ctx):
    """heart.Covariance interface + multivariate_normal_chol(datasets, weights, hyperparams,
    residuals) adaptor against the reference golden / scipy (test_models.py:149-222)"""
    import scipy.stats
    from types import SimpleNamespace

    from beat_amd.heart import Covariance, chol_inverse_batch, log_determinant
    from beat_amd.models import multivariate_normal_chol
    g = load_golden("covariance")
    for k in g["names"]:
        c = Covariance(data=g[k + "_C"])
        np.testing.assert_allclose(c.chol_inverse, g[k + "_W"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(c.log_pdet, g[k + "_logpdet"], rtol=1e-13)
        np.testing.assert_allclose(c.inverse(), g[k + "_inv"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(float(c.slog_pdet.get_value()), g[k + "_logpdet"], rtol=1e-13)
        np.testing.assert_allclose(log_determinant(g[k + "_C"]), g[k + "_logdet_fn"], rtol=1e-13)
    ct = Covariance(data=g["toeplitz_C"], pred_v=0.1 * np.eye(48))
    np.testing.assert_allclose(ct.chol_inverse, g["total_W"], rtol=1e-10, atol=1e-12)
    # batched factorisation on the GPU: W^T W == inv(C) (test_covariance.py:71-112 criterion)
    Cs = np.stack([g["toeplitz_C"] * s for s in (0.5, 1.0, 2.0)])
    W, lp = chol_inverse_batch(Cs)
    for i in range(3):
        np.testing.assert_allclose(W[i].T @ W[i], np.linalg.inv(Cs[i]), rtol=0, atol=1e-6)
        np.testing.assert_allclose(lp[i], Covariance(data=Cs[i]).log_pdet, rtol=1e-12)
    # adaptor: 2 datasets x 10 samples, C = 0.001 I, hyper 0 -> scipy logpdf
    rng = np.random.default_rng(1)
    Cd = 0.001 * np.eye(10)
    dsets = [SimpleNamespace(typ="any_P_T", samples=10, covariance=Covariance(data=Cd)) for _ in range(2)]
    weights = [d.covariance.chol_inverse for d in dsets]
    res = 0.03 * rng.standard_normal((2, 10))
    out = multivariate_normal_chol(dsets, weights, {"h_any_P_T": 0.0}, res)
    for i in range(2):
        np.testing.assert_allclose(out[i], scipy.stats.multivariate_normal.logpdf(res[i], np.zeros(10), Cd),
                                   rtol=0, atol=1e-6)
    outb = multivariate_normal_chol(dsets, weights, {"h_any_P_T": np.array([[0.0, 0.3]] * 4)},
                                    np.stack([res] * 4), hp_specific=True)
    assert outb.shape == (4, 2) and np.allclose(outb[:, 0], out[0])


@pytest.mark.parametrize("covariance,N", [("scalar", 128), ("scalar", 100), ("toeplitz", 192)])
@pytest.mark.parametrize("C", [512, 700, 1100])
def test_float_storage_kernel_equals_the_f64_kernels_on_the_rounded_library(ctx, monkeypatch, covariance, N, C):
    """SURVEY 8(f) row 2 "optional fp32 layout": ``LogpForwFunc.round_libraries_to_f32`` rounds the libraries in
    place (explicit, irreversible) and keeps float copies of the same values; the 512-chain-group kernel reads
    those (k_gfstack_ws32: 256-byte row segments, operands widened before the f64 FMA).  Because both
    copies hold the same numbers the result equals -- bit for bit -- the f64 loader/consumer kernel on the
    rounded library, the streaming kernel to 1e-12 and the oracle run on float-rounded G at 1e-9."""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((4,), (5,), (1.0,), T=3, N=N, D=3, S=25, covariance=covariance)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    monkeypatch.setenv("BEATAMD_GS_CG", "512")    # (tiny problems may measure a smaller group as faster)
    before = f.batch(Q)
    with pytest.raises(ValueError):
        f.set_f32(True)          # nothing rounds a library implicitly
    f.round_libraries_to_f32()
    L32 = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws32<"), ctx.last_kernel()
    assert not np.array_equal(L32, before)      # the library was rounded: 1e-8-level changes
    # float storage is NOT inside north_star's 1e-6 on every problem (a labelled option, never `value`): the
    # misfit of a chain near the data is a small difference of large synthetics.  Here: 1e-5 on `like`
    np.testing.assert_allclose(L32[:, -1], before[:, -1], rtol=1e-5)
    f.set_f32(False)
    L64 = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws<"), ctx.last_kernel()
    assert np.array_equal(L32, L64)
    monkeypatch.setenv("BEATAMD_GS_PAIR", "1")            # the same pair gather on the float64 rows (ds_read_b128)
    Lp = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_wsp64<"), ctx.last_kernel()
    assert np.array_equal(Lp, L64)
    monkeypatch.delenv("BEATAMD_GS_PAIR")
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    Ls = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<"), ctx.last_kernel()
    np.testing.assert_allclose(Ls, L32, rtol=1e-12)     # (other tile sums of the misfit: not bitwise)
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    f.set_f32(True)
    monkeypatch.delenv("BEATAMD_GS_CG")
    sub = f.batch(np.ascontiguousarray(Q[100:164]))      # a small batch: no float kernel, the equal f64 values
    assert not ctx.last_kernel().startswith("k_gfstack_ws32") and np.array_equal(sub, L32[100:164])
    host32 = dict(host)
    host32["Gs"] = [g.astype(np.float32).astype(np.float64) for g in host["Gs"]]
    for c in (0, C // 2, C - 1):
        ref, _ = problem_oracle.forward(host32, Q[c])
        np.testing.assert_allclose(L32[c], ref, rtol=1e-9)


def test_float_copy_is_dropped_when_the_rows_are_rewritten(ctx, monkeypatch):
    """the float copy of a rounded library must never outlive its float64 rows.  An upload into the
    library and an in-place whitening (``beatamd_whiten_rows``, what a covariance update of a pre-whitened model
    does) drop the copy and take the model off it -- the likelihoods are those of the NEW float64 rows, through
    the float64 kernel; float storage is refused for a pre-whitened wavemap and is never switched on implicitly."""
    import torch
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((4,), (5,), (1.0,), T=3, N=128, D=3, S=25)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 512)
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    f.round_libraries_to_f32()
    L32 = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws32<"), ctx.last_kernel()
    gf = prob.wavemaps[0].gfs["uparr"]
    # (1) new values uploaded into the library
    G2 = host["Gs"][0] * 1.5
    ctx.seis_gflib_upload(gf.lib_id, np.ascontiguousarray(G2).reshape(-1), 0)
    L2 = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws<"), ctx.last_kernel()     # off the (dropped) float copy
    host2 = dict(host)
    host2["Gs"] = [G2]
    from oracle import problem_oracle
    for c in (0, 300, 511):
        ref, _ = problem_oracle.forward(host2, Q[c])
        np.testing.assert_allclose(L2[c], ref, rtol=1e-9)
    with pytest.raises(Exception):
        f.set_f32(True)         # the copy is gone: asking for it fails loudly (round again to get one)
    # (2) rows rewritten in place by a whitening (what a covariance update of a pre-whitened model does)
    prob_d, host_d = build_problem(spec, device_library=True, ctx=ctx)
    f_d = prob_d.compile(ctx)
    f_d.round_libraries_to_f32()
    assert f_d.batch(Q) is not None and ctx.last_kernel().startswith("k_gfstack_ws32<"), ctx.last_kernel()
    gd = prob_d.wavemaps[0].gfs["uparr"]
    N = spec.N
    W = np.triu(np.random.default_rng(1).standard_normal((N, N)) * 0.01) + np.eye(N)
    G_rounded = gd._device_tensor.cpu().numpy().copy()
    ctx.whiten_rows(gd._device_tensor.view(-1, N), W)
    L3 = f_d.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws<"), ctx.last_kernel()
    host3 = dict(host_d)
    host3["Gs"] = [G_rounded @ W.T]
    for c in (0, 300, 511):
        ref, _ = problem_oracle.forward(host3, Q[c])
        np.testing.assert_allclose(L3[c], ref, rtol=1e-9)
    # (3) a pre-whitened wavemap refuses float storage
    spec_t = SyntheticSpec((4,), (5,), (1.0,), T=3, N=64, D=3, S=25, covariance="toeplitz")
    prob_t, _ = build_problem(spec_t)
    f_t = prob_t.compile(ctx, prewhiten=True)
    with pytest.raises(ValueError):
        f_t.round_libraries_to_f32()


@pytest.mark.parametrize("sizes", [(1,), (16,), (17, 33), (64, 5, 100), (257,), (512,), (300, 3, 31, 130), (513, 20)])
@pytest.mark.parametrize("C", [1, 15, 16, 40])
def test_small_dense_geodetic_datasets_in_one_launch(ctx, sizes, C):
    """k_quadform_small: all dense geodetic datasets of a composite (M <= 512 each) in one launch with
    the MVN epilogue -- sizes around the 16-row tiles, the 32-column prefetch batches, the 64 KB LDS
    boundary (M = 512), several datasets, chain counts around the 16-chain blocks; a dataset above 512
    points sends the composite through the tiled k_quadform path.  Against the oracle composition."""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((3,), (4,), (1.0,), T=0, N=0, D=3, S=25, geodetic_nobs=sizes, seed=sum(sizes) + C)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    LL = f.batch(Q)
    assert LL.shape == (C, len(sizes) + 1)
    for c in sorted({0, C // 2, C - 1}):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(LL[c], ref, rtol=1e-6)      # north_star tolerance
        np.testing.assert_allclose(LL[c], ref, rtol=1e-10, atol=1e-9)
    # a chain's value does not depend on the batch it is evaluated in
    if C > 1:
        assert np.array_equal(f.batch(np.ascontiguousarray(Q[C - 1:])), LL[C - 1:])


@pytest.mark.parametrize("interp,covariance,C", [("multilinear", "scalar", 512), ("multilinear", "toeplitz", 300),
                                                  ("multilinear", "scalar", 100), ("nearest_neighbor", "scalar", 300),
                                                  ("nearest_neighbor", "toeplitz", 130)])
def test_float_storage_lds_dma_kernel(ctx, monkeypatch, interp, covariance, C):
    """k_gfstack_dmaf: the LDS-DMA kernel on the float copies (multilinear: the four-row blend of the
    reference's default interpolation; nearest neighbour in groups below 512 chains) -- bit for bit the
    float64 kernels (cell kernel / k_gfstack_dma) on the rounded library, oracle on float-rounded G at 1e-9."""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((4,), (5,), (1.0,), T=3, N=136, D=3, S=25, covariance=covariance, interpolation=interp)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    f.round_libraries_to_f32()
    L32 = f.batch(Q)
    # (300 nearest-neighbour chains run as one 512-chain group: the loader/consumer float kernel)
    assert ctx.last_kernel().startswith(("k_gfstack_dmaf<", "k_gfstack_ws32<")), ctx.last_kernel()
    assert interp != "multilinear" or ctx.last_kernel().startswith("k_gfstack_dmaf<")
    f.set_f32(False)
    L64 = f.batch(Q)
    assert not ctx.last_kernel().startswith(("k_gfstack_dmaf", "k_gfstack_ws32")), ctx.last_kernel()
    assert np.array_equal(L32, L64), ctx.last_kernel()
    host32 = dict(host)
    host32["Gs"] = [g.astype(np.float32).astype(np.float64) for g in host["Gs"]]
    for c in (0, C // 2, C - 1):
        ref, _ = problem_oracle.forward(host32, Q[c])
        np.testing.assert_allclose(L32[c], ref, rtol=1e-9)


@pytest.mark.parametrize("M", [40, 300, 1030])
def test_banded_whitening_operators(ctx, orc, monkeypatch, M):
    """the reference's "exponential" noise structure (covariance.py:24-51) is a Markov kernel: W = chol(inv(C)).T is
    bidiagonal up to rounding residue -> the weight set is evaluated on its band (k_quadform_banded).  Against the dense
    kernel (BEATAMD_QF_BAND=0), the oracle's dense product, and: wider bands, a band too wide, a non-triangular matrix,
    small matrices, update from banded to dense and back"""
    rng = np.random.default_rng(M)
    nd, C = 3, 37
    Ws, slogs = [], []
    for d in range(nd):
        Cd = (0.3 + d) * orc.exponential_data_covariance(M, 0.5, 2.0)
        Ws.append(orc.cov_chol_inverse(Cd))
        slogs.append(orc.cov_log_pdet(Cd))
    W = np.stack(Ws)
    assert np.abs(np.triu(W, 2)).max() < 1e-13 * np.abs(W).max() and np.abs(np.triu(W, 1)).max() > 0.1 * np.abs(W).max()
    wid = ctx.weights_create_dense(W, slogs)
    assert ctx.weights_band(wid) == 1
    res = rng.standard_normal((C, nd, M))
    hp = rng.uniform(-1, 1, (C, nd))
    out = ctx.mvn_chol_logp_batch(wid, res, hp)
    monkeypatch.setenv("BEATAMD_QF_BAND", "0")
    assert ctx.weights_band(wid) == -1
    dense = ctx.mvn_chol_logp_batch(wid, res, hp)
    monkeypatch.delenv("BEATAMD_QF_BAND")
    np.testing.assert_allclose(out, dense, rtol=1e-10)
    assert np.array_equal(out, ctx.mvn_chol_logp_batch(wid, res, hp))          # (fixed summation order)
    for c in range(0, C, 9):
        np.testing.assert_allclose(out[c], orc.multivariate_normal_chol(Ws, slogs, hp[c], res[c]), rtol=1e-10)
    # a band of five (an upper-triangular operator built directly), one dataset with a narrower band
    Wb = np.zeros((nd, M, M))
    for d in range(nd):
        for k in range(6 if d else 3):
            Wb[d] += np.diag(rng.uniform(0.5, 1.5, M - k) * (0.5 ** k), k)
    ctx.weights_update(wid, Wb, slogs)
    assert ctx.weights_band(wid) == 5
    out5 = ctx.mvn_chol_logp_batch(wid, res, hp)
    for c in range(0, C, 9):
        np.testing.assert_allclose(out5[c], orc.multivariate_normal_chol(list(Wb), slogs, hp[c], res[c]), rtol=1e-10)
    # an entry far from the diagonal, just above / below the threshold: 2^-40 of the largest entry OF ITS ROW (round 6,
    # a matrix-wide maximum would drop entries that matter in a row of small scale)
    far = Wb.copy()
    far[1, 0, M - 1] = 2.0 ** -39 * np.abs(Wb[1, 0]).max()
    ctx.weights_update(wid, far, slogs)
    assert ctx.weights_band(wid) == (-1 if M > 33 else 5)
    far[1, 0, M - 1] = 2.0 ** -41 * np.abs(Wb[1, 0]).max()
    ctx.weights_update(wid, far, slogs)
    assert ctx.weights_band(wid) == 5
    band, dropped = ctx.weights_band_info(wid)
    assert band == 5 and dropped == 2.0 ** -41          # what the banded evaluation leaves out, relative to the row
    # heterogeneous row scales: a row a million times smaller keeps an entry that is large IN THAT ROW
    het = Wb.copy()
    het[1, 3] *= 1e-6
    het[1, 3, M - 2] = 0.3 * np.abs(het[1, 3]).max()
    ctx.weights_update(wid, het, slogs)
    assert ctx.weights_band(wid) == -1
    np.testing.assert_allclose(ctx.mvn_chol_logp_batch(wid, res, hp)[::9],
                               [orc.multivariate_normal_chol(list(het), slogs, hp[c], res[c]) for c in range(0, C, 9)], rtol=1e-10)
    # below the diagonal: not triangular -> dense
    low = Wb.copy()
    low[2, 5, 4] = 1e-300
    ctx.weights_update(wid, low, slogs)
    assert ctx.weights_band(wid) == -1
    np.testing.assert_allclose(ctx.mvn_chol_logp_batch(wid, res, hp)[::9],
                               [orc.multivariate_normal_chol(list(low), slogs, hp[c], res[c]) for c in range(0, C, 9)], rtol=1e-10)
    # NaN in an operator: dense (and NaN out, as the reference)
    bad = Wb.copy()
    bad[0, 3, 3] = np.nan
    ctx.weights_update(wid, bad, slogs)
    assert ctx.weights_band(wid) == -1
    ctx.weights_update(wid, W, slogs)
    assert ctx.weights_band(wid) == 1
    assert np.array_equal(ctx.mvn_chol_logp_batch(wid, res, hp), out)
    ctx.weights_destroy(wid)
    # small matrices stay with the dense kernel
    wid = ctx.weights_create_dense(W[:, :20, :20].copy(), slogs)
    assert ctx.weights_band(wid) == -1
    ctx.weights_destroy(wid)


@pytest.mark.parametrize("N,C,nvar,shifts", [(64, 300, 1, False), (200, 300, 1, True), (130, 1100, 1, False), (192, 257, 2, True),
                                             (1030, 64, 1, False)])
def test_bidiagonal_misfit_inside_the_stacking_kernel(ctx, monkeypatch, N, C, nvar, shifts):
    """round 6: with the reference's "exponential" noise structure (covariance.py:24-51; bidiagonal
    W = chol(inv(C)).T) the misfit sum_i (W[i,i] r_i + W[i,i+1] r_{i+1})^2 (distributions.py:119-138) rides in the
    epilogue of k_gfstack_ws -- no residual store, no second kernel: the inner samples of a 64-sample tile in the kernel,
    every tile's last sample (its neighbour is the next tile's first residual) in the tile-sum kernel.  Against the
    unfused path (residual store + k_quadform_banded, BEATAMD_QF_FUSE=0), the dense kernel (BEATAMD_QF_BAND=0) and the
    oracle's dense product; single tile, ragged last tile (N = 200, 130, 1030), several chain groups, two slip
    components, station shifts; the same bits on every call"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    names = ("uparr", "uperp")[:nvar]
    spec = SyntheticSpec((6,), (7,), (1.0,), T=3, N=N, D=3, S=25, covariance="toeplitz", slip_varnames=names,
                         station_shifts=shifts)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    assert ctx.weights_band(f.problem.wavemaps[0]._wset) == 1
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    monkeypatch.setenv("BEATAMD_GS_CG", "512")      # (a library this small would be stacked in smaller groups)
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws<1,3,"), ctx.last_kernel()
    assert np.array_equal(A, f.batch(Q))
    monkeypatch.setenv("BEATAMD_QF_FUSE", "0")
    B = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_ws<1,2,"), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_QF_FUSE")
    # ONE summation order for every path (quadform.hip k_quadform_band1): a chain's misfit has the same bits whichever kernel
    # stacked it -- batch size, group size and rank count cannot show
    assert np.array_equal(A, B)
    monkeypatch.setenv("BEATAMD_QF_BAND", "0")
    D = f.batch(Q)
    monkeypatch.delenv("BEATAMD_QF_BAND")
    np.testing.assert_allclose(A, D, rtol=1e-10)
    # kernels without the epilogue: the caller still gets the banded misfit (residual store + k_quadform_banded)
    monkeypatch.setenv("BEATAMD_GS_CG", "128")
    E = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_dma<2,1,2,"), ctx.last_kernel()
    assert np.array_equal(A, E)
    assert np.array_equal(A[40:90], f.batch(Q[40:90]))      # a sub-batch: another kernel, the same bits
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    S_ = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<"), ctx.last_kernel()
    if nvar == 1:
        assert np.array_equal(A, S_)          # (one slip variable: the chain-shared kernels are bitwise the streaming kernel)
    np.testing.assert_allclose(A, S_, rtol=1e-11)
    for c in (0, C // 2, C - 1):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(A[c], ref, rtol=1e-9)
    f.release()
