"""Randomised cross-checks of the stacking kernels (GPU): every chain-group kernel against the
streaming kernel on random library shapes, batch sizes, interpolation modes and epilogues.  The
kernels accumulate a chain's rows in the same order with the same fma, so synthetics are compared
bit for bit; fused log-likelihoods (tile sums in a different order) to 1e-11."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import beat_amd
    return beat_amd.get_context(0)


def _lib(ctx, G, st_dt, du_dt):
    from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    T, P, D, S, N = G.shape
    gf = SeismicGFLibrary(SeismicGFLibraryConfig(dimensions=G.shape, starttime_sampling=st_dt,
                                                 duration_sampling=du_dt, starttime_min=0.0, duration_min=0.5))
    gf.setup(T, P, D, S, N, allocate=True)
    gf._gfmatrix[:] = G
    gf.init_optimization(ctx)
    return gf


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_synthetics_bitwise(ctx, monkeypatch, seed):
    """stack_all_batch: random (T, P, D, S, N, chains), nn and multilinear, every group size and
    both 512-chain kernels against the streaming kernel"""
    rng = np.random.default_rng(100 + seed)
    T, P = int(rng.integers(1, 5)), int(rng.integers(1, 30))
    D, S = int(rng.integers(1, 4)), int(rng.integers(2, 40))
    N = 2 * int(rng.integers(1, 200))
    C = int(rng.choice([48, 63, 64, 65, 129, 300, 513, 700, 1100, 1500]))
    G = rng.standard_normal((T, P, D, S, N))
    gf = _lib(ctx, G, 0.5, 0.5)
    dur = rng.uniform(0.5, 0.5 + 0.5 * (D - 1) + 0.2, (C, P))
    hi = 0.5 * (S - 1)
    st = rng.uniform(0.0, hi * rng.uniform(0.2, 1.0), (C, T, P))   # narrow or wide row spread
    sl = rng.uniform(0, 5, (C, P))
    for interp in ("nearest_neighbor", "multilinear"):
        if interp == "multilinear" and (D < 2 or S < 2):
            continue
        d_ = np.clip(dur, 0.5 + 1e-9, 0.5 + 0.5 * (D - 1)) if interp == "multilinear" else dur
        s_ = np.clip(st, 1e-9, hi) if interp == "multilinear" else st
        monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
        ref = gf.stack_all_batch(d_, s_, sl, interpolation=interp)
        assert ctx.last_kernel().startswith("k_gfstack<")
        monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
        monkeypatch.delenv("BEATAMD_GS_CG", raising=False)
        out = gf.stack_all_batch(d_, s_, sl, interpolation=interp)      # the measured group size
        assert np.array_equal(out, ref), (seed, interp, "tuned", ctx.last_kernel())
        out = gf.stack_all_batch(d_, s_, sl, interpolation=interp)      # and again (fitted twin launches)
        assert np.array_equal(out, ref), (seed, interp, "tuned again", ctx.last_kernel())
        for cg in ("64", "128", "256", "512"):
            monkeypatch.setenv("BEATAMD_GS_CG", cg)
            for ws in ("1", "0"):
                monkeypatch.setenv("BEATAMD_GS_WS", ws)
                out = gf.stack_all_batch(d_, s_, sl, interpolation=interp)
                assert np.array_equal(out, ref), (seed, interp, cg, ws, ctx.last_kernel())
                if cg != "512":
                    break
        monkeypatch.delenv("BEATAMD_GS_WS")


@pytest.mark.parametrize("seed", range(12))
def test_random_fused_models(ctx, monkeypatch, seed):
    """f.batch on random problem shapes (scalar / dense covariance = both misfit epilogues, station
    shifts on / off = per-target or shared index tables, one or two slip variables), many chain
    groups; chain-group kernels against the streaming kernel"""
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    rng = np.random.default_rng(500 + seed)
    nd, ns = int(rng.integers(2, 7)), int(rng.integers(2, 7))
    spec = SyntheticSpec((nd,), (ns,), (1.0,), T=int(rng.integers(1, 7)), N=2 * int(rng.integers(8, 150)),
                         D=3, S=25, covariance=str(rng.choice(["scalar", "toeplitz"])),
                         station_shifts=bool(rng.integers(0, 2)),
                         slip_varnames=("uparr", "uperp")[:int(rng.integers(1, 3))],
                         interpolation=str(rng.choice(["nearest_neighbor", "multilinear"])))
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = int(rng.choice([100, 530, 2100, 5000]))
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    A = f.batch(Q)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "1")
    tag = (seed, spec.covariance, spec.station_shifts, spec.interpolation, len(spec.slip_varnames), C)
    for cg in (None, "64", "128", "256", "512"):
        if cg is None:
            monkeypatch.delenv("BEATAMD_GS_CG", raising=False)
        else:
            monkeypatch.setenv("BEATAMD_GS_CG", cg)
        for rep in range(2):
            B = f.batch(Q)
            np.testing.assert_allclose(A, B, rtol=1e-11, atol=1e-9, err_msg=str(tag + (cg, ctx.last_kernel())))
