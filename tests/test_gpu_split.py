"""Patch split of small-N libraries (round 6): a library of short traces (the reference's realistic case,
SURVEY 8(d) config 4: 120 samples) has T * ceil(N / 64) (target, tile) walks of P serial steps each -- 70 for 256 CUs.  Such
libraries are stacked as the view [T*R, P/R, D, S, N] of the same memory (R patch ranges per target: R x as many, R x shorter
walks), the ranges' partial synthetics summed in range order (gfstack.hip launch_gfstack_split).  R depends on the library's
shape only, every stacking kernel takes the view unchanged.  Reference arithmetic: beat/ffi/base.py:607-709 (stack_all),
beat/models/seismic.py:1283-1349."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import beat_amd
    return beat_amd.get_context(0)


@pytest.mark.parametrize("interp", ["nearest_neighbor", "multilinear"])
@pytest.mark.parametrize("cov,shifts,nvar", [("scalar", False, 1), ("toeplitz", True, 2), ("scalar", True, 1)])
def test_small_n_library_is_stacked_in_patch_ranges(ctx, monkeypatch, interp, cov, shifts, nvar):
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    names = ("uparr", "uperp")[:nvar]
    # 128 patches, 3 targets x 120 samples: 6 walks -> R = 4 ranges of 32 patches (the rule wants >= 32 patches per range)
    spec = SyntheticSpec((8,), (16,), (1.0,), T=3, N=120, D=3, S=25, covariance=cov, slip_varnames=names,
                         station_shifts=shifts, interpolation=interp, st_dt=0.5)
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    C = 600
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], C)
    A = f.batch(Q)
    kern = ctx.last_kernel()
    plan = ctx.gf_plan()["plan"]
    assert kern.startswith("k_gfstack_ws<" if interp == "nearest_neighbor" else "k_gfstack_runs<"), (kern, plan)
    assert "stacked in 4 ranges of 32" in plan, plan
    assert np.isfinite(A).all() and np.array_equal(A, f.batch(Q))
    # the same library unsplit (BEATAMD_GF_SPLIT=0): another summation order over the patches, same numbers to rounding
    monkeypatch.setenv("BEATAMD_GF_SPLIT", "0")
    U = f.batch(Q)
    assert "ranges" not in ctx.gf_plan()["plan"]
    monkeypatch.delenv("BEATAMD_GF_SPLIT")
    np.testing.assert_allclose(A, U, rtol=1e-10)
    # a chain's result does not depend on its batch: sub-batches through the same and through other kernels
    sub = f.batch(Q[40:140])                                 # (100 chains: the k_gfstack_dma family on the same view)
    if nvar == 1 or interp == "nearest_neighbor":
        assert np.array_equal(A[40:140], sub)
    np.testing.assert_allclose(A[40:140], sub, rtol=1e-11)    # two slip variables: the kernels interleave them differently
    small = f.batch(Q[7:27])                                  # 20 chains: the streaming kernel, on the same view
    assert ctx.last_kernel().startswith("k_gfstack<"), ctx.last_kernel()
    if nvar == 1:
        assert np.array_equal(A[7:27], small)
    np.testing.assert_allclose(A[7:27], small, rtol=1e-11)
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    S_ = f.batch(Q)
    assert ctx.last_kernel().startswith("k_gfstack<") and "ranges" in ctx.gf_plan()["plan"]
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    if nvar == 1:
        assert np.array_equal(A, S_)
    np.testing.assert_allclose(A, S_, rtol=1e-11)
    # another number of ranges on request: R = 2, 8 (16 patches per range)
    for R in ("2", "8"):
        monkeypatch.setenv("BEATAMD_GF_SPLIT", R)
        np.testing.assert_allclose(f.batch(Q), A, rtol=1e-10)
        assert "stacked in %s ranges" % R in ctx.gf_plan()["plan"]
    monkeypatch.delenv("BEATAMD_GF_SPLIT")
    for c in (0, C // 2, C - 1):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(A[c], ref, rtol=1e-9)
    # out-of-library chains are still flagged (NaN like + IndexError at the next synchronisation)
    Qb = Q[:64].copy()
    Qb[5, host["layout"].offset("durations") + 100] = 99.0      # a patch of the LAST range
    with pytest.raises(IndexError):
        f.batch(Qb)                                            # host arrays: checked before the results are handed back
    import torch
    Lb = f.batch(torch.from_numpy(Qb).to(torch.device("cuda", 0))).cpu().numpy()
    with pytest.raises(IndexError):
        ctx.synchronize()
    assert np.isnan(Lb[5, -1]) and np.isfinite(np.delete(Lb[:, -1], 5)).all()
    f.release()


def test_synthetics_of_a_split_library_equal_the_oracle(ctx):
    """stack_all through the library API (beat/ffi/base.py:607-709) on a small-N library: explicit start times [C, T, P]"""
    from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    from oracle import oracle as orc
    T, P, D, S, N = 4, 96, 3, 20, 100
    rng = np.random.default_rng(3)
    G = rng.standard_normal((T, P, D, S, N))
    cfg = SeismicGFLibraryConfig(dimensions=(T, P, D, S, N), starttime_sampling=0.5, duration_sampling=0.5, starttime_min=0.0,
                                 duration_min=0.5)
    gf = SeismicGFLibrary(cfg)
    gf.setup(T, P, D, S, N, allocate=False)
    gf._gfmatrix = G
    gf.init_optimization(ctx)
    C = 70
    dur = rng.uniform(0.5, 1.5, (C, P))
    st = rng.uniform(0.0, 9.0, (C, T, P))
    sl = rng.uniform(0, 3, (C, P))
    for interp in ("nearest_neighbor", "multilinear"):
        out = gf.stack_all_batch(dur, st, sl, interpolation=interp)
        assert "stacked in 3 ranges of 32" in ctx.gf_plan()["plan"], ctx.gf_plan()
        for c in (0, 33, C - 1):
            ref = orc.stack_all(G, dur[c], st[c], sl[c], 0.5, 0.5, 0.0, 0.5, interpolation=interp)
            np.testing.assert_allclose(out[c], ref, rtol=1e-11, atol=1e-11)
