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
    ctx.synchronize()
