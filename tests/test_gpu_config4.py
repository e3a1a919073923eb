"""BASELINE configs[3] at its stress size and (round 6, second test) at the realistic trace length of 120 samples (SURVEY
8(d) "config 4": the shape of the reference's
test_ffi_gfstacking_multifault.py -- 2 subfaults of 10x20 patches, 35 targets, station time shifts,
two slip components -- with N = 4096 samples and the per-GPU share of 4096 chains over 8 GPUs = 512
chains), joint with the geodetic composite on the real Laquila SAR scenes.  The two seismic
libraries (2 x 55 GB) are generated in HBM; the oracle check runs on sampled (chain, target) pairs
with the rows of those targets copied to the host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("interp,kernel", [("multilinear", "k_gfstack_runs<3,"), ("nearest_neighbor", "k_gfstack_ws<1,3,3,")])
def test_config4_joint_multifault_512_chains_n4096(interp, kernel):
    """both interpolations (the reference's default for this composite is multilinear, beat/config.py:571-575); the
    kernel that stacks the 512-chain batch is asserted by name: the runs kernel / the loader-consumer kernel, both
    outside their round-4 envelopes on this (2 durations x 60 start times) library; both carry the bidiagonal misfit of the
    Toeplitz covariance in their epilogues (mode 3, round 6), the 64-chain sub-batch below goes through another kernel +
    k_quadform_band1 and must give the same bits"""
    import torch

    import beat_amd
    from beat_amd.models.problem import GeodeticData
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from conftest import load_golden
    from oracle import oracle as orc
    ctx = beat_amd.get_context(0)
    free, _ = torch.cuda.mem_get_info(0)
    if free < 140e9:
        pytest.skip("needs 140 GB of free HBM")
    g = load_golden("laquila_geodetic")
    sizes = tuple(int(g["d%d_displacement" % i].size) for i in range(int(g["n"])))
    T, N, D, S = 35, 4096, 2, 60
    spec = SyntheticSpec((10, 10), (20, 20), (2.0, 2.0), T=T, N=N, D=D, S=S, st_dt=0.5,
                         slip_varnames=("uparr", "uperp"), covariance="toeplitz", station_shifts=True,
                         geodetic_nobs=sizes, vel_bounds=(3.0, 4.0), time_bounds=(0.0, 2.0), interpolation=interp)
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    gd = prob.geodetic
    data = np.concatenate([g["d%d_displacement" % i] for i in range(len(sizes))])
    odw = np.concatenate([g["d%d_odw" % i] for i in range(len(sizes))])
    Ws = [orc.cov_chol_inverse(g["d%d_C" % i]) for i in range(len(sizes))]
    sl = [float(g["d%d_logpdet" % i]) for i in range(len(sizes))]
    prob.geodetic = GeodeticData(gd.gfs, data, odw, sizes, Ws, sl, gd.hypers)
    host.update(gdata=data, godw=odw, gW=Ws, gslog=sl)
    f = prob.compile(ctx)
    lay = host["layout"]
    C = 512
    Q = draw_population(spec, lay, host["lower"], host["upper"], C)
    Qd = torch.from_numpy(Q).cuda()
    LL = f.batch(Qd)
    ctx.synchronize()
    assert ctx.last_kernel().startswith(kernel), (ctx.last_kernel(), ctx.gf_plan())
    kernel = ctx.last_kernel()
    LL = LL.cpu().numpy()
    assert LL.shape == (C, T + 2 + 1) and np.isfinite(LL).all()
    np.testing.assert_allclose(LL[:, -1], LL[:, :T].sum(1) + LL[:, T:T + 2].sum(1), rtol=1e-12)
    # a batch is a set of independent chains: any sub-batch gives the same rows (other group sizes,
    # other kernels), to the last bit
    sub = f.batch(Qd[37:101].contiguous()).cpu().numpy()
    assert np.array_equal(sub, LL[37:101]), (kernel, ctx.last_kernel())
    # oracle composition on sampled targets: rows of those targets copied to the host
    tsel = np.array([0, 17, 34])
    Gs_sub = [prob.wavemaps[0].gfs[v]._device_tensor[torch.from_numpy(tsel).cuda()].cpu().numpy()
              for v in spec.slip_varnames]
    name, sidx = host["time_shifts"]
    for c in (0, 255, 511):
        pt = lay.rmap(Q[c])
        slips = np.stack([pt[v] for v in spec.slip_varnames])
        hp = np.array([pt["h_any_P_0_Z"][i] for _, i in host["hypers"]])[tsel]
        ts = pt[name][np.asarray(sidx)][tsel]
        _, _, logpts = orc.ffi_seismic_forward(
            Gs_sub, dict(dur_min=spec.du_min, dur_dt=spec.du_dt, st_min=spec.st_min, st_dt=spec.st_dt),
            dict(ndip=spec.n_patch_dip, nstrike=spec.n_patch_strike, patch_size=spec.patch_size),
            dict(slips=slips, durations=pt["durations"], velocities=pt["velocities"],
                 nuc_strike=pt["nucleation_strike"], nuc_dip=pt["nucleation_dip"], time=pt["time"]),
            host["data"][tsel], host["weights"][tsel], host["slog"][tsel], hp, time_shifts=ts,
            interpolation=spec.interpolation)
        np.testing.assert_allclose(LL[c, tsel], logpts, rtol=1e-6)   # north_star tolerance
        np.testing.assert_allclose(LL[c, tsel], logpts, rtol=1e-9)
        hps = [pt["h_SAR"][i] for _, i in host["ghyp"]]
        lg, _ = orc.ffi_geodetic_logp(host["gGs"], slips, data, odw, sizes, Ws, sl, hps)
        np.testing.assert_allclose(LL[c, T:T + 2], lg, rtol=1e-9)
    # one Metropolis step of the whole population through the fused step call keeps the bookkeeping
    lo, up = lay.bounds(host["lower"], host["upper"])
    from beat_amd.sampler.metropolis import BatchedMetropolis
    st = BatchedMetropolis(f, lo, up, C, device=torch.device("cuda", 0), seed=3)
    st.set_proposal(np.diag(((up - lo) * 2e-3) ** 2))
    L0 = torch.from_numpy(LL).cuda()
    n_acc = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(2):
        st.step(Qd, L0, 0.05, n_acc)
    ctx.synchronize()
    assert 0 < int(n_acc.item()) <= 2 * C and int(st.accepted_since_tune.sum().item()) == int(n_acc.item())
    np.testing.assert_allclose(f.batch(Qd).cpu().numpy(), L0.cpu().numpy(), rtol=1e-11, atol=1e-9)
    del Gs_sub


@pytest.mark.parametrize("interp,kernel", [("multilinear", "k_gfstack_runs<0,"), ("nearest_neighbor", "k_gfstack_ws<1,0,3,")])
def test_config4_joint_multifault_512_chains_n120(interp, kernel):
    """BASELINE configs[3] at the REALISTIC trace length (120 samples = 60 s at 2 Hz; round 6): the
    libraries (2 x 1.6 GB) live on the host too, so the oracle composes the whole model for sampled chains.  The launch is
    asserted through the plan: 10 patch ranges of 40 = 700 walks instead of 70 for the 256 CUs, index tables per station
    slot; every chain of the batch equals its value in other batches / through other kernels to rounding (the ranges'
    partial synthetics are summed in one fixed order per library), and the Metropolis step keeps its books"""
    import torch

    import beat_amd
    from beat_amd.models.problem import GeodeticData
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from conftest import load_golden
    from oracle import oracle as orc
    from oracle import problem_oracle
    ctx = beat_amd.get_context(0)
    g = load_golden("laquila_geodetic")
    sizes = tuple(int(g["d%d_displacement" % i].size) for i in range(int(g["n"])))
    T, N, D, S = 35, 120, 2, 60
    spec = SyntheticSpec((10, 10), (20, 20), (2.0, 2.0), T=T, N=N, D=D, S=S, st_dt=0.5,
                         slip_varnames=("uparr", "uperp"), covariance="toeplitz", station_shifts=True,
                         geodetic_nobs=sizes, vel_bounds=(3.0, 4.0), time_bounds=(0.0, 2.0), interpolation=interp)
    prob, host = build_problem(spec)
    gd = prob.geodetic
    data = np.concatenate([g["d%d_displacement" % i] for i in range(len(sizes))])
    odw = np.concatenate([g["d%d_odw" % i] for i in range(len(sizes))])
    Ws = [orc.cov_chol_inverse(g["d%d_C" % i]) for i in range(len(sizes))]
    sl = [float(g["d%d_logpdet" % i]) for i in range(len(sizes))]
    prob.geodetic = GeodeticData(gd.gfs, data, odw, sizes, Ws, sl, gd.hypers)
    host.update(gdata=data, godw=odw, gW=Ws, gslog=sl)
    f = prob.compile(ctx)
    lay = host["layout"]
    C = 512
    Q = draw_population(spec, lay, host["lower"], host["upper"], C)
    Qd = torch.from_numpy(Q).cuda()
    LL = f.batch(Qd)
    ctx.synchronize()
    plan = ctx.gf_plan()
    assert ctx.last_kernel().startswith(kernel), (ctx.last_kernel(), plan)
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert "stacked in 10 ranges of 40 (700 walks instead of 70)" in plan["plan"], plan
    LL = LL.cpu().numpy()
    assert LL.shape == (C, T + 2 + 1) and np.isfinite(LL).all()
    np.testing.assert_allclose(LL[:, -1], LL[:, :T].sum(1) + LL[:, T:T + 2].sum(1), rtol=1e-12)
    assert np.array_equal(LL, f.batch(Qd).cpu().numpy())
    sub = f.batch(Qd[37:101].contiguous()).cpu().numpy()          # (64 chains: another kernel on the same ranges)
    np.testing.assert_allclose(sub, LL[37:101], rtol=1e-11)
    assert np.array_equal(sub[:, T:T + 2], LL[37:101, T:T + 2])   # the geodetic datasets: bitwise whatever the batch
    for c in (0, 255, 511):
        ref, _ = problem_oracle.forward(host, Q[c])
        np.testing.assert_allclose(LL[c], ref, rtol=1e-6)         # north_star tolerance
        np.testing.assert_allclose(LL[c], ref, rtol=1e-9)
    lo, up = lay.bounds(host["lower"], host["upper"])
    from beat_amd.sampler.metropolis import BatchedMetropolis
    st = BatchedMetropolis(f, lo, up, C, device=torch.device("cuda", 0), seed=3)
    st.set_proposal(np.diag(((up - lo) * 2e-3) ** 2))
    L0 = torch.from_numpy(LL).cuda()
    n_acc = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(3):
        st.step(Qd, L0, 0.05, n_acc)
    ctx.synchronize()
    assert 0 < int(n_acc.item()) <= 3 * C and int(st.accepted_since_tune.sum().item()) == int(n_acc.item())
    np.testing.assert_allclose(f.batch(Qd).cpu().numpy(), L0.cpu().numpy(), rtol=1e-11, atol=1e-9)
    f.release()
