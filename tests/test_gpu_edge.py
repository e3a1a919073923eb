"""Edge cases of the hot path on the GPU: empty and ragged batches, degenerate grids, extreme
library shapes, boundary indices -- each against the oracle."""
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


def _lib(ctx, G, st_min=0.0, st_dt=0.5, du_min=0.5, du_dt=0.5):
    from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    gf = SeismicGFLibrary(SeismicGFLibraryConfig(dimensions=G.shape, starttime_sampling=st_dt,
                                                 duration_sampling=du_dt, starttime_min=st_min,
                                                 duration_min=du_min))
    gf.setup(*G.shape, allocate=True)
    gf._gfmatrix[:] = G
    gf.init_optimization(ctx)
    return gf


def test_empty_batches(ctx):
    gf = _lib(ctx, np.ones((2, 3, 1, 1, 4)))
    out = gf.stack_all_batch(np.zeros((0, 3)), np.zeros((0, 2, 3)), np.zeros((0, 3)))
    assert out.shape == (0, 2, 4)
    assert ctx.fast_sweep_batch(np.zeros((0, 6)), 1.0, np.zeros(0, np.int32), np.zeros(0, np.int32),
                                2, 3).shape == (0, 6)
    from beat_amd.synthetic import SyntheticSpec, build_problem
    spec = SyntheticSpec((3,), (3,), (1.0,), T=2, N=8, D=2, S=12)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    assert f.batch(np.zeros((0, host["layout"].size))).shape == (0, f.nllk)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1, 2), (1, 1, 1, 1, 1), (2, 1, 2, 3, 7), (1, 9, 1, 1, 3),
                                   (3, 2, 1, 4, 514), (1, 2, 2, 2, 4096)])
@pytest.mark.parametrize("C", [1, 3, 65])
def test_degenerate_library_shapes(ctx, orc, monkeypatch, shape, C):
    T, P, D, S, N = shape
    rng = np.random.default_rng(sum(shape) + C)
    G = rng.standard_normal(shape)
    gf = _lib(ctx, G)
    dur = 0.5 + 0.5 * rng.uniform(0, D - 1, (C, P)) if D > 1 else np.full((C, P), 0.5)
    st = 0.5 * rng.uniform(0, S - 1, (C, T, P)) if S > 1 else np.zeros((C, T, P))
    sl = rng.uniform(-2, 2, (C, P))
    for kern in ("0", "1"):
        monkeypatch.setenv("BEATAMD_GF_KERNEL", kern)
        for interp in ("nearest_neighbor", "multilinear"):
            if interp == "multilinear" and (D == 1 or S == 1) and N % 2 == 0:
                # a one-node axis: ceil-1 wraps to the same node (numpy semantics), still defined
                pass
            out = gf.stack_all_batch(dur, st, sl, interpolation=interp)
            for c in (0, C - 1):
                ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.5, 0.0, 0.5, interp)
                np.testing.assert_allclose(out[c], ref, rtol=1e-11, atol=1e-12)


def test_boundary_indices(ctx, orc):
    """times exactly on the last node, on half-way ties, and a hair outside the grid"""
    T, P, D, S, N = 1, 6, 3, 5, 8
    G = np.random.default_rng(0).standard_normal((T, P, D, S, N))
    gf = _lib(ctx, G)
    dur = np.array([[0.5, 1.0, 1.5, 0.75, 1.25, 1.5]])            # nodes, ties (round-half-even)
    st = np.array([[[0.0, 2.0, 0.25, 0.75, 1.25, 1.75]]])          # 0.25/.75 -> ties at x.5
    sl = np.ones((1, P))
    for interp in ("nearest_neighbor", "multilinear"):
        out = gf.stack_all_batch(dur, st, sl, interpolation=interp)[0]
        ref = orc.stack_all(G, dur[0], st[0], sl[0], 0.5, 0.5, 0.0, 0.5, interp)
        np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-13)
    st_bad = st.copy()
    st_bad[0, 0, 3] = 2.26  # rint(4.52) = 5 -> outside S = 5
    with pytest.raises(IndexError):
        gf.stack_all_batch(dur, st_bad, sl)
    st_neg = st.copy()
    st_neg[0, 0, 3] = -0.4  # rint(-0.8) = -1 -> numpy wraps to the last node
    out = gf.stack_all_batch(dur, st_neg, sl)[0]
    ref = orc.stack_all(G, dur[0], st_neg[0], sl[0], 0.5, 0.5, 0.0, 0.5)
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("nd,ns", [(1, 1), (1, 5), (5, 1), (2, 2), (80, 80), (3, 100)])
def test_sweep_degenerate_and_large_grids(ctx, orc, nd, ns):
    rng = np.random.default_rng(nd * 100 + ns)
    C = 5
    slow = 1.0 / rng.uniform(1.0, 5.0, (C, nd * ns))
    hd = rng.integers(0, nd, C).astype(np.int32)
    hs = rng.integers(0, ns, C).astype(np.int32)
    out = ctx.fast_sweep_batch(slow, 0.7, hd, hs, nd, ns)
    for c in range(C):
        np.testing.assert_allclose(out[c], orc.fast_sweep(slow[c], 0.7, hd[c], hs[c], nd, ns),
                                   rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        ctx.fast_sweep_batch(np.ones((1, 81 * 81)), 1.0, [0], [0], 81, 81)  # beyond the LDS limit


def test_nan_and_inf_flow_like_the_reference(ctx, orc):
    """fast_sweep_ext.c control flow on NaN/inf compares: an infinitely slow patch
    is by-passed or stays +inf exactly as in the C"""
    nd, ns = 4, 5
    slow = np.full((2, nd * ns), 0.3)
    slow[0, 7] = np.inf
    slow[1, :] = np.inf
    slow[1, 0] = 0.3
    out = ctx.fast_sweep_batch(slow, 1.0, np.array([0, 0], np.int32), np.array([0, 0], np.int32), nd, ns)
    for c in range(2):
        ref = orc.fast_sweep(slow[c], 1.0, 0, 0, nd, ns)
        assert np.array_equal(np.isfinite(out[c]), np.isfinite(ref))
        m = np.isfinite(ref)
        np.testing.assert_allclose(out[c][m], ref[m], rtol=0, atol=1e-12)


def test_weights_update_and_reuse(ctx, orc):
    """seismic.py:1509-1534 update_weights: new chol_inverse / slog_pdet take effect"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((4,), (4,), (1.0,), T=3, N=24, D=3, S=25, covariance="toeplitz")
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 5)
    a = f.batch(Q)
    Cn = np.stack([(1.5 + t) * orc.exponential_data_covariance(24, 0.5, 3.0) for t in range(3)])
    Wn = np.stack([orc.cov_chol_inverse(c) for c in Cn])
    sn = np.array([orc.cov_log_pdet(c) for c in Cn])
    f.update_weights(0, Wn, sn)
    host["weights"], host["slog"] = Wn, sn
    b = f.batch(Q)
    assert not np.allclose(a, b)
    for c in range(5):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(b[c], ref, rtol=1e-9)


@pytest.mark.parametrize("D,S,C,interp,kernels,passes", [
    (12, 20, 300, "multilinear", ("k_gfstack_runs<0,",), True),            # 252 dense slots per patch: row passes
    (12, 20, 300, "nearest_neighbor", ("k_gfstack_ws<1,0,3,", "k_gfstack_dma<"), None),   # whichever group size measures fastest
    (40, 64, 130, "multilinear", ("k_gfstack<1,",), False),                # below 192 chains, no two row buffers: streaming
    (3, 11, 600, "nearest_neighbor", ("k_gfstack_ws<1,0,3,", "k_gfstack_dma<"), None),
    (17, 41, 600, "nearest_neighbor", ("k_gfstack_ws<1,0,3,", "k_gfstack_dma<"), None),   # the tutorial grid
    (17, 41, 600, "multilinear", ("k_gfstack_runs<0,",), True),
    (40, 64, 600, "multilinear", ("k_gfstack_runs<0,",), True),            # ~1700 slots per patch: the tables overflow
    (3, 25, 600, "multilinear", ("k_gfstack_runs<0,",), False),            # config 3: one pass per patch
])
def test_kernel_selection_by_lds_capacity(ctx, orc, monkeypatch, D, S, C, interp, kernels, passes):
    """libraries with many (duration, start-time) rows: which kernel the selection takes (by NAME) -- the row-pass
    kernels whatever the grid (k_gfstack_ws for one row per chain, k_gfstack_runs for multilinear from 192 chains on),
    k_gfstack_dma where two row buffers of a small group's bound fit LDS, else the streaming kernel; a population that
    needs more row passes than the runs kernel's tables hold is stacked by the streaming kernel standing in.  Every
    choice gives the streaming kernel's bits and the oracle's values"""
    T, P, N = 2, 5, 64
    rng = np.random.default_rng(D * S + C)
    G = rng.standard_normal((T, P, D, S, N))
    gf = _lib(ctx, G)
    dur = 0.5 + 0.5 * rng.uniform(0, D - 1, (C, P))
    st = 0.5 * rng.uniform(0, S - 1, (C, T, P))
    sl = rng.uniform(-2, 2, (C, P))
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    a = gf.stack_all_batch(dur, st, sl, interpolation=interp)
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    b = gf.stack_all_batch(dur, st, sl, interpolation=interp)   # default selection
    name, plan = ctx.last_kernel(), ctx.gf_plan()
    assert name.startswith(kernels), (name, plan)
    if passes is not None:
        assert (plan["max_passes"] >= 2) == passes, plan
    assert plan["plan"], plan
    assert np.array_equal(a, b)
    if name.startswith("k_gfstack_ws") and D * S > 96:
        assert plan["max_passes"] >= 2, plan
    for c in (0, C // 3, C - 1):
        ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.5, 0.0, 0.5, interp)
        np.testing.assert_allclose(b[c], ref, rtol=1e-11, atol=1e-12)


def test_release_frees_the_models_device_state(ctx):
    """LogpForwFunc.release(): the model record and its weight sets (T x N x N doubles each when dense) are destroyed, the
    libraries stay; the function refuses further calls, a second compile of the same problem works"""
    import torch
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((4,), (4,), (1.0,), T=3, N=512, D=3, S=25, covariance="toeplitz", geodetic_nobs=(9,))
    prob, host = build_problem(spec)
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 6)
    f = prob.compile(ctx)
    a = f.batch(Q)
    free0, _ = torch.cuda.mem_get_info(0)
    f.release()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 - free0 >= 3 * 512 * 512 * 8 * 0.9          # the dense weight set went back
    with pytest.raises((ValueError, TypeError, RuntimeError)):
        f.batch(Q)
    f.release()                                              # idempotent
    g = prob.compile(ctx)
    assert np.array_equal(g.batch(Q), a)
