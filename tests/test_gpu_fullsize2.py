"""Full-size parity, two slip variables: T=64, P=400, D=3, S=25, N=4096 with TWO 62.9 GB libraries
(uparr, uperp) resident in HBM (125.8 GB) -- the step of k_gfstack_dma alternates between the
libraries (800 steps per workgroup).  Separate module so that the one-library fixture of
test_gpu_fullsize.py is released first."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T, P, D, S, N = 64, 400, 3, 25, 4096


def test_fullsize_two_slip_variables_dma_vs_streaming_vs_rows(monkeypatch):
    import torch

    import beat_amd
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import oracle as orc
    import gc
    ctx = beat_amd.get_context(0)
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(0)
    if free < 150e9:
        pytest.skip("needs 150 GB of free HBM")
    spec = SyntheticSpec((20,), (20,), (1.0,), T=T, N=N, D=D, S=S, slip_varnames=("uparr", "uperp"),
                         time_bounds=(0.0, 0.0))
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    f = prob.compile(ctx)
    lay = host["layout"]
    C = 530
    Q = draw_population(spec, lay, host["lower"], host["upper"], C, seed_offset=31000)
    Qd = torch.from_numpy(Q).to("cuda:0")
    monkeypatch.delenv("BEATAMD_GF_KERNEL", raising=False)
    monkeypatch.setenv("BEATAMD_GS_CG", "512")
    LL = f.batch(Qd).cpu().numpy()
    assert ctx.last_kernel().startswith("k_gfstack_ws<1,1,3,"), ctx.last_kernel()   # the default for 512-chain groups
    monkeypatch.setenv("BEATAMD_GS_WS", "0")
    LD = f.batch(Qd).cpu().numpy()
    assert ctx.last_kernel().startswith("k_gfstack_dma<8,1,1,64,"), ctx.last_kernel()
    assert np.array_equal(LL, LD)
    monkeypatch.delenv("BEATAMD_GS_WS")
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    LS = f.batch(Qd).cpu().numpy()
    assert ctx.last_kernel().startswith("k_gfstack<0,2,"), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    np.testing.assert_allclose(LL, LS, rtol=1e-12)
    # sampled (chain, target): synthetics from the rows themselves (fetched from HBM by index)
    Gs = [prob.wavemaps[0].gfs[v]._device_tensor for v in ("uparr", "uperp")]
    for c in (0, 529):
        pt = lay.rmap(Q[c])
        hd, hs = orc.positions2idxs([pt["nucleation_dip"][0], pt["nucleation_strike"][0]], 1.0)
        st0 = orc.fast_sweep(1.0 / pt["velocities"], 1.0, int(hd), int(hs), 20, 20) + pt["time"][0]
        di, _ = orc.time2idx(pt["durations"], spec.du_min, spec.du_dt)
        si, _ = orc.time2idx(st0, spec.st_min, spec.st_dt)
        pi = torch.arange(P, device="cuda:0")
        dit, sit = (torch.from_numpy(a.astype(np.int64)).to("cuda:0") for a in (di, si))
        for t in (5, 63):
            syn = np.zeros(N)
            for G, v in zip(Gs, ("uparr", "uperp")):
                rows = G[t, pi, dit, sit, :].cpu().numpy()
                syn += (rows * pt[v][:, None]).sum(0)
            ref = orc.mvn_chol_logp(host["weights"][t], host["data"][t] - syn, host["slog"][t],
                                    pt["h_any_P_0_Z"][0])
            np.testing.assert_allclose(LL[c, t], ref, rtol=1e-10)
    # ---- the reference's default interpolation on the same two libraries: k_gfstack_runs with 800 steps per
    # workgroup (the steps cycle through the variables' libraries patch by patch)
    from beat_amd.models.problem import FFIProblem, SeismicWavemap
    wm0 = prob.wavemaps[0]
    wm = SeismicWavemap(wm0.gfs, wm0.data, wm0.weights, wm0.slog_pdet, wm0.hypers, wm0.time_shifts, "multilinear")
    fm = FFIProblem(prob.layout, prob.n_patch_dip, prob.n_patch_strike, prob.patch_sizes, prob.slip_varnames,
                    [wm], None, None, prob.lower, prob.upper).compile(ctx)
    monkeypatch.delenv("BEATAMD_GS_CG")
    LM = fm.batch(Qd).cpu().numpy()
    assert ctx.last_kernel().startswith("k_gfstack_runs<1,"), ctx.last_kernel()
    monkeypatch.setenv("BEATAMD_GF_KERNEL", "0")
    LMS = fm.batch(Qd).cpu().numpy()
    assert ctx.last_kernel().startswith("k_gfstack<1,2,"), ctx.last_kernel()
    monkeypatch.delenv("BEATAMD_GF_KERNEL")
    np.testing.assert_allclose(LM, LMS, rtol=1e-11)
    for c in (0, 529):
        pt = lay.rmap(Q[c])
        hd, hs = orc.positions2idxs([pt["nucleation_dip"][0], pt["nucleation_strike"][0]], 1.0)
        st0 = orc.fast_sweep(1.0 / pt["velocities"], 1.0, int(hd), int(hs), 20, 20) + pt["time"][0]
        di, df = orc.time2idx(pt["durations"], spec.du_min, spec.du_dt, "multilinear")
        si, sf = orc.time2idx(st0, spec.st_min, spec.st_dt, "multilinear")
        pi = torch.arange(P, device="cuda:0")
        for t in (5, 63):
            syn = np.zeros(N)
            for dd, ss, w in ((di, si, (1 - sf) * (1 - df)), (di, si - 1, sf * (1.0 - df)),
                              (di - 1, si, (1 - sf) * df), (di - 1, si - 1, sf * df)):
                dd = np.where(dd < 0, dd + D, dd); ss = np.where(ss < 0, ss + S, ss)
                dit, sit = (torch.from_numpy(a.astype(np.int64)).to("cuda:0") for a in (dd, ss))
                for G, v in zip(Gs, ("uparr", "uperp")):
                    rows = G[t, pi, dit, sit, :].cpu().numpy()
                    syn += (rows * (w * pt[v])[:, None]).sum(0)
            ref = orc.mvn_chol_logp(host["weights"][t], host["data"][t] - syn, host["slog"][t],
                                    pt["h_any_P_0_Z"][0])
            np.testing.assert_allclose(LM[c, t], ref, rtol=1e-9)
    ctx.synchronize()
