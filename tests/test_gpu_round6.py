"""Round 6's scheduling changes leave every number where it was: index tables built by one wavefront per patch
(k_gm_tables_w) against the workgroup-per-patch kernel, the geodetic stack's blocks of four chains against single chains.  Reference arithmetic: beat/ffi/base.py:292-305,
607-709; beat/models/geodetic.py:1065-1081; beat/models/problems.py:227-247."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import beat_amd
    return beat_amd.get_context(0)


@pytest.mark.parametrize("D,S,cap", [(2, 60, None), (17, 41, None), (3, 25, "10"), (5, 25, "8")])
def test_tables_by_wavefront_equal_tables_by_workgroup(ctx, monkeypatch, D, S, cap):
    """multilinear, two slip variables, station shifts, a population over the whole grid: row passes along the duration axis
    (122 / 714 dense slots per patch against 104 row slots; buffers of 10 / 8 slots cut duration lines along the start-time
    axis too) -- the runs kernel on the tables of either builder gives the same bits, and the streaming kernel's values"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    spec = SyntheticSpec((6,), (8,), (1.5,), T=5, N=200, D=D, S=S, slip_varnames=("uparr", "uperp"), station_shifts=True,
                         interpolation="multilinear", st_dt=0.5, du_min=0.0 if D > 3 else 0.5, du_dt=0.25 if D > 3 else 0.5,
                         covariance="toeplitz")
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = 700
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    if cap:
        monkeypatch.setenv("BEATAMD_GR_CAP", cap)
        monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "40")
    monkeypatch.setenv("BEATAMD_GF_SPLIT", "0")                   # (the library as it is: one walk per target and tile)
    monkeypatch.setenv("BEATAMD_GM_WAVE", "1")                    # (by default only from 16 patches per CU on)
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<"), (ctx.last_kernel(), ctx.gf_plan())
    plan_w = ctx.gf_plan()
    monkeypatch.setenv("BEATAMD_GM_WAVE", "0")
    B = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<")
    plan_g = ctx.gf_plan()
    monkeypatch.delenv("BEATAMD_GM_WAVE")
    assert np.isfinite(A).all() and np.array_equal(A, B)
    assert plan_w["mean_passes"] == plan_g["mean_passes"] and plan_w["max_passes"] == plan_g["max_passes"], (plan_w, plan_g)
    if cap or D * (S + 1) > 300:         # (2 x 61 dense slots: the count phase runs, a patch rarely needs a second pass)
        assert plan_w["max_passes"] >= 2, plan_w
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    S_ = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<1,")
    np.testing.assert_allclose(A, S_, rtol=1e-11)                   # (two variables: the kernels interleave them differently)
    f.release()


@pytest.mark.parametrize("nvar", [1, 2])
def test_geodetic_stack_in_blocks_of_four_chains(ctx, nvar):
    """ffi/base.py:292-305 batched: 70 chains go through the 4-chain blocks (one launch for all slip variables), 10-chain
    slices through the one-chain kernel: the same fma sequence per (chain, point) -- bitwise; numpy at rounding"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    names = ("uparr", "uperp")[:nvar]
    spec = SyntheticSpec((3,), (7,), (2.0,), T=2, N=64, D=2, S=25, slip_varnames=names, geodetic_nobs=(130, 33))
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = 70
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    A = f.batch(Q)
    for a in range(0, C, 10):
        sub = f.batch(Q[a:a + 10])
        assert np.array_equal(A[a:a + 10, 2:4], sub[:, 2:4]), a         # the two geodetic datasets' columns
        np.testing.assert_allclose(A[a:a + 10], sub, rtol=1e-11)
    # the library method itself against numpy
    from beat_amd.ffi import GeodeticGFLibrary, GeodeticGFLibraryConfig
    rng = np.random.default_rng(5)
    P, Nobs = 37, 301
    G = rng.standard_normal((P, Nobs))
    gl = GeodeticGFLibrary(GeodeticGFLibraryConfig(dimensions=(P, Nobs)))
    gl.setup(P, Nobs, allocate=False)
    gl._gfmatrix = G
    gl.init_optimization(ctx)
    sl = rng.uniform(-1, 4, (C, P))
    mu = gl.stack_all_batch(sl)
    np.testing.assert_allclose(mu, sl @ G, rtol=1e-12, atol=1e-12)
    assert np.array_equal(mu[:9], gl.stack_all_batch(sl[:9]))
    f.release()


@pytest.mark.parametrize("N,nvar,shifts", [(130, 1, False), (200, 2, True), (64, 1, True)])
def test_bidiagonal_misfit_inside_the_runs_kernel(ctx, monkeypatch, N, nvar, shifts):
    """multilinear interpolation with the reference's "exponential" noise structure (bidiagonal W = chol(inv(C)).T,
    covariance.py:24-51): the misfit of distributions.py:119-138 rides in the epilogue of k_gfstack_runs too (mode 3 of the
    generated program: the transposed tile of the scalar epilogue, the band rows of the tile in LDS, edges + k_sum_tiles_band1
    for every tile's last sample) -- in the one canonical order, so the same bits as residual store + k_quadform_band1
    (BEATAMD_QF_FUSE=0), as the small-batch kernels and, when the tables overflow, as the streaming stand-in with its guarded
    k_quadform_band1; row passes included"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    names = ("uparr", "uperp")[:nvar]
    spec = SyntheticSpec((6,), (7,), (1.0,), T=3, N=N, D=3, S=25, covariance="toeplitz", slip_varnames=names,
                         station_shifts=shifts, interpolation="multilinear")
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    assert ctx.weights_band(f.problem.wavemaps[0]._wset) == 1
    C = 600
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    monkeypatch.setenv("BEATAMD_GF_SPLIT", "0")
    A = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<3,"), (ctx.last_kernel(), ctx.gf_plan())
    assert np.isfinite(A).all() and np.array_equal(A, f.batch(Q))
    monkeypatch.setenv("BEATAMD_QF_FUSE", "0")
    B = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<2,"), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_QF_FUSE")
    assert np.array_equal(A, B)
    # row passes (buffers of 10 slots), the same tables through both builders
    monkeypatch.setenv("BEATAMD_GR_CAP", "10")
    monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "40")
    P_ = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<3,") and ctx.gf_plan()["max_passes"] >= 2, ctx.gf_plan()
    assert np.array_equal(A, P_)
    # tables sized for one pass per patch overflow: the streaming kernel stands in, its residuals go through the guarded
    # k_quadform_band1
    monkeypatch.setenv("BEATAMD_GR_PASS_ALLOC", "1")
    O = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack_runs<3,")
    if nvar == 1:
        assert np.array_equal(A, O)
    np.testing.assert_allclose(A, O, rtol=1e-11)
    monkeypatch.delenv("BEATAMD_GR_CAP")
    monkeypatch.delenv("BEATAMD_GR_PASS_ALLOC")
    sub = f.batch(Q[40:140])                                   # 100 chains: the lane <-> chain kernels + k_quadform_band1
    if nvar == 1:
        assert np.array_equal(A[40:140], sub)
    np.testing.assert_allclose(A[40:140], sub, rtol=1e-11)
    monkeypatch.setenv("BEATAMD_QF_BAND", "0")                 # the dense operator on the matrix cores
    D_ = f.batch(Q)
    monkeypatch.delenv("BEATAMD_QF_BAND")
    np.testing.assert_allclose(A, D_, rtol=1e-10)
    for c in (0, C // 2, C - 1):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(A[c], ref, rtol=1e-9)
    f.release()
