"""CPU tests of the sampler drivers (host logic) on toy targets, following the reference's
test/test_smc.py (two-Gaussian mixture, mean |x| = 0.5 +- 0.03) and the decision rules of
metropolis.py / pt.py pinned through the oracle."""
import numpy as np
import pytest

from conftest import load_golden


def _two_gaussians(n=4, stdev=0.1):
    """reference test/test_smc.py:38-66"""
    mu1 = np.ones(n) * (1.0 / 2)
    mu2 = -mu1
    sigma = np.power(stdev, 2) * np.eye(n)
    isigma = np.linalg.inv(sigma)
    dsigma = np.linalg.det(sigma)
    w1, w2 = stdev, 1 - stdev

    def f(X):
        d1, d2 = X - mu1, X - mu2
        c = -0.5 * n * np.log(2 * np.pi) - 0.5 * np.log(dsigma)
        l1 = c - 0.5 * np.einsum("ci,ij,cj->c", d1, isigma, d1)
        l2 = c - 0.5 * np.einsum("ci,ij,cj->c", d2, isigma, d2)
        return np.log(w1 * np.exp(l1) + w2 * np.exp(l2))
    return f, n


def test_smc_two_gaussians():
    from beat_amd.sampler import SMC, smc_sample
    from beat_amd.sampler.hosttarget import HostTarget
    f, n = _two_gaussians()
    step = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=300, tune_interval=10,
               random_seed=3)
    pop, lp, betas = smc_sample(100, step)
    assert betas[0] == 0.0 and betas[-1] == 1.0 and all(np.diff(betas) >= 0)
    # reference test/test_smc.py:112-115
    np.testing.assert_allclose(np.abs(pop).mean(axis=0), 0.5, rtol=0, atol=0.03)
    assert lp.shape == (300, 1)


def test_host_stage_ops_match_reference_golden():
    """calc_beta / resample / weighted covariance of the host back end (toy targets) against arrays
    captured from the reference's SMC methods (oracle/gen_golden.py); the device back end is
    checked against the same fixtures in tests/test_gpu_samplers.py"""
    import torch
    from beat_amd.sampler.ops import HostOps
    ops = HostOps()
    g = load_golden("smc")
    for k in range(int(g["ncase"])):
        lk = torch.from_numpy(g["c%d_lk" % k])
        b, w = ops.calc_beta(lk, float(g["c%d_beta_in" % k]), 1.0)
        assert b == float(g["c%d_beta" % k])
        np.testing.assert_array_equal(w.numpy(), g["c%d_w" % k])
        idx = ops.resample(w, float(np.ravel(g["c%d_aux" % k])[0]))
        assert np.array_equal(idx.numpy(), g["c%d_idx" % k])
        F = ops.population_factor(torch.from_numpy(g["c%d_pop" % k]), w).numpy()
        np.testing.assert_allclose(F.T @ F, g["c%d_cov" % k], rtol=1e-12, atol=1e-15)


def test_tune_tables():
    from beat_amd.sampler import pt, smc, step_tune
    g = load_golden("smc")
    for a, p, s in zip(g["tune_acc"], g["pt_tune"], g["smc_tune"]):
        assert pt.tune(1.2, a) == p and smc.tune(a) == s
    np.testing.assert_allclose(step_tune(np.ones(6), [0.0005, 0.03, 0.1, 0.3, 0.8, 0.99]),
                               [0.1, 0.5, 0.9, 1.0, 2.0, 10.0])


def test_philox_reference_known_answers():
    """the Python twin of the device generator (tests/philox_ref.py) reproduces the published
    Philox4x32-10 known-answer vectors (Random123 kat_vectors)"""
    from philox_ref import philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(x[0]) for x in got) == want


def test_pt_manager_and_toy_sampling():
    from beat_amd.sampler import TemperingManager, pt_sample
    from beat_amd.sampler.hosttarget import HostTarget
    from oracle import oracle as orc
    man = TemperingManager(2, 4, n_replicas=3)
    # pt.py:200-203 ladder
    np.testing.assert_allclose(man.betas, [1, 1, 1 / 1.2, 1 / 1.2 ** 2, 1 / 1.2 ** 3, 1 / 1.2 ** 4])
    assert man.chain_betas.shape == (18,)
    like = np.linspace(-30, 0, 18)
    perm = man.swap_round(like)
    assert sorted(perm.tolist()) == list(range(18))  # a permutation
    # every swap obeys the reference rule for the draw that was used
    man2 = TemperingManager(2, 4, n_replicas=3)
    logu = np.log(np.random.RandomState(17).uniform(size=(6, 3)))
    l2 = like.reshape(6, 3)
    for k in (0, 2, 4):
        for r in range(3):
            want = orc.pt_swap_accept(man2.betas[k], man2.betas[k + 1], l2[k, r], l2[k + 1, r], logu[k, r])
            assert (perm[k * 3 + r] == (k + 1) * 3 + r) == want
    assert 100 <= man.draw_swap_interval() < 300
    # toy target: two Gaussians, posterior replicas must find both modes' |x|
    f, n = _two_gaussians()
    s, ls, man = pt_sample(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains_posterior=2,
                           n_chains_tempered=6, n_replicas=8, n_samples=1500, swap_interval=(20, 40),
                           beta_tune_interval=5, proposal_cov=np.eye(n) * 0.02, random_seed=5)
    assert s.shape == (1500, n)
    np.testing.assert_allclose(np.abs(s[300:]).mean(axis=0), 0.5, rtol=0, atol=0.06)
    assert 1.01 <= man.current_scale <= 2.0 and len(man.history) > 0


def test_smc_stage_traces_and_resume(tmp_path):
    """stage directories in the reference's trace format + resume from a completed stage"""
    import os
    from collections import OrderedDict

    from beat_amd.backend import NumpyChain, stage_path
    from beat_amd.models import ParameterLayout
    from beat_amd.sampler import SMC, smc_sample
    from beat_amd.sampler.hosttarget import HostTarget
    f, n = _two_gaussians()
    lay = ParameterLayout(OrderedDict([("x", n)]))
    step = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=64, tune_interval=10,
               random_seed=3)
    pop, lp, betas = smc_sample(20, step, homepath=str(tmp_path), layout=lay, out_names=["like"])
    nstage = len(betas) - 2
    assert os.path.isdir(stage_path(str(tmp_path), 0)) and os.path.isdir(stage_path(str(tmp_path), -1))
    ch = NumpyChain.load(os.path.join(stage_path(str(tmp_path), -1), "chain-5.bin"))
    np.testing.assert_array_equal(ch.get_values("x")[0], pop[5])
    assert ch.get_values("like")[0] == lp[5, 0]
    # resume after stage 1 with a sampler seeded differently: the stage state (population, beta,
    # per-chain step sizes and tuning counters, the shared random stream) is restored, so the
    # resumed run repeats the uninterrupted one bit for bit
    step2 = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=64, tune_interval=10,
                random_seed=4)
    pop2, lp2, betas2 = smc_sample(20, step2, homepath=str(tmp_path), layout=None, resume_stage=1)
    assert nstage >= 2
    assert betas2 == betas[1:]
    assert np.array_equal(pop2, pop) and np.array_equal(lp2, lp)
    # buffer_thinning: the stage files hold every 6th draw of every chain and the last (draws 0, 6, 12, 18, 19) -- the
    # reference's traces (beat/backend.py:365-404) -- and the last record is the chain's end point; the same run otherwise
    step3 = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=64, tune_interval=10, random_seed=3)
    home3 = str(tmp_path / "thinned")
    pop3, lp3, betas3 = smc_sample(20, step3, homepath=home3, layout=lay, out_names=["like"], buffer_thinning=6)
    assert betas3 == betas and np.array_equal(pop3, pop) and np.array_equal(lp3, lp)
    ch = NumpyChain.load(os.path.join(stage_path(home3, -1), "chain-5.bin"))
    assert ch.get_values("x").shape == (5, n)
    np.testing.assert_array_equal(ch.get_values("x")[-1], pop[5])
    assert ch.get_values("like")[-1] == lp[5, 0]
    one = NumpyChain.load(os.path.join(stage_path(home3, 0), "chain-5.bin"))
    assert one.get_values("x").shape == (1, n)             # (stage 0: one evaluation, no move)
    with pytest.raises(ValueError, match="buffer_thinning writes trace files"):
        smc_sample(4, step3, buffer_thinning=2)


def test_host_proposal_draws_normal_and_cauchy_statistics():
    """host back end of the proposal draw (the device back end is tested on the GPU): rows z . F
    have covariance F^T F (MultivariateNormal), resp. are multivariate t with one degree of freedom
    (MultivariateCauchy, base.py:35-71,163-186: z / sqrt(chi2(1)))"""
    import torch
    from beat_amd.sampler.base import covariance_factor, proposal_df
    from beat_amd.sampler.ops import HostOps
    ops = HostOps()
    rng = np.random.default_rng(3)
    A = rng.standard_normal((4, 4))
    cov = A @ A.T + 0.5 * np.eye(4)
    F = torch.from_numpy(covariance_factor(cov))
    np.testing.assert_allclose(F.numpy().T @ F.numpy(), cov, rtol=1e-12)
    x, lu = ops.draw(F, 200000, seed=5, step=0)
    np.testing.assert_allclose(np.cov(x.numpy().T), cov, rtol=0.03, atol=0.03)
    assert (lu.numpy() < 0).all() and abs(np.exp(lu.numpy()).mean() - 0.5) < 0.01
    y, _ = ops.draw(F, 200000, seed=6, step=0, df=proposal_df("MultivariateCauchy"))
    y = y.numpy()
    # marginals of a multivariate Cauchy are Cauchy with scale sqrt(cov_ii): median |y_i| = scale
    np.testing.assert_allclose(np.median(np.abs(y), axis=0), np.sqrt(np.diag(cov)), rtol=0.03)
    # heavy tails: P(|y| > 10 scale) = 1 - 2/pi atan(10) = 0.0635
    frac = (np.abs(y[:, 0]) > 10 * np.sqrt(cov[0, 0])).mean()
    assert abs(frac - 0.0635) < 0.004
    # different steps / chains give different rows; the same (seed, step, chain) the same rows
    a, _ = ops.draw(F, 8, seed=1, step=3)
    b, _ = ops.draw(F, 8, seed=1, step=3)
    c, _ = ops.draw(F, 8, seed=1, step=4)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # a singular covariance (population smaller than the parameter count) is repaired, not rejected
    B = rng.standard_normal((6, 2))
    Fs = covariance_factor(B @ B.T)
    np.testing.assert_allclose(Fs.T @ Fs, B @ B.T, atol=1e-10)
    # the per-parameter families (base.py:129-155) are drawn component by component, no factor
    assert proposal_df("Normal") is None and proposal_df("Laplace") is None and proposal_df("Poisson") is None
    with pytest.raises(NotImplementedError):
        proposal_df("Binomial")
    with pytest.raises(ValueError):
        covariance_factor(np.array([[np.nan]]))
    # host twin of the per-parameter draws: Laplace has variance 2 scale^2, Cauchy quartiles at +-scale
    from beat_amd.sampler.ops import HostOps
    ops = HostOps()
    sc = torch.tensor([0.5, 2.0], dtype=torch.float64)
    lap, lu = ops.draw_univariate(2, sc, 100000, seed=4, step=1)
    assert np.allclose(lap.numpy().var(axis=0), 2.0 * sc.numpy() ** 2, rtol=0.05) and lu.shape == (100000,)
    cau, _ = ops.draw_univariate(1, sc, 100000, seed=4, step=2)
    assert np.allclose(np.quantile(cau.numpy(), 0.75, axis=0), sc.numpy(), rtol=0.05)
    poi, _ = ops.draw_univariate(3, sc, 100000, seed=4, step=3)       # poisson(lam = scale) - scale (base.py:150-155)
    assert np.allclose(poi.numpy().var(axis=0), sc.numpy(), rtol=0.05) and np.all(np.abs(poi.numpy().mean(axis=0)) < 0.02)
    # a Metropolis stepper with a per-parameter proposal samples the toy target
    from beat_amd.sampler import SMC, smc_sample
    from beat_amd.sampler.hosttarget import HostTarget
    f, n = _two_gaussians()
    step = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=200, tune_interval=10, random_seed=8,
               proposal_name="Normal", scale=0.1)
    pop, lp, betas = smc_sample(40, step)
    assert np.allclose(np.abs(pop).mean(axis=0), 0.5, atol=0.08), np.abs(pop).mean(axis=0)


def test_population_factor_equals_weighted_covariance():
    """population_factor reproduces np.cov(X, aweights=w) (SMC.calc_covariance) as F^T F without
    forming it, also when the population is smaller than the parameter count"""
    import torch
    from beat_amd.sampler.ops import HostOps
    ops = HostOps()
    rng = np.random.default_rng(11)
    for n, d in ((40, 6), (5, 9)):
        X = rng.standard_normal((n, d)) * rng.uniform(0.5, 3.0, d) + rng.standard_normal(d)
        w = rng.random(n)
        w /= w.sum()
        cov = np.cov(X, aweights=w, bias=False, rowvar=0)
        F = ops.population_factor(torch.from_numpy(X), torch.from_numpy(w))
        np.testing.assert_allclose(F.numpy().T @ F.numpy(), cov, rtol=1e-12, atol=1e-12)
        rows, _ = ops.draw(F, 300000, seed=2, step=0)
        np.testing.assert_allclose(np.cov(rows.numpy().T), cov, rtol=0.04, atol=0.04 * np.abs(cov).max())


def test_degenerate_weights_raise_like_calc_covariance():
    """one chain carries all the importance weight: v1 - v2/v1 = 0, the weighted sample covariance
    is 0/0 -- the reference's calc_covariance raises (smc.py:181-185); a non-finite population
    likewise (the device path used to continue with a NaN factor and a frozen population)"""
    import torch
    from beat_amd.sampler.ops import HostOps
    ops = HostOps()
    X = torch.from_numpy(np.random.default_rng(0).standard_normal((50, 4)))
    w = torch.zeros(50, dtype=torch.float64)
    w[7] = 1.0
    with pytest.raises(ValueError, match="Sample covariances contains Inf or NaN"):
        ops.population_factor(X, w)
    w = torch.full((50,), 0.02, dtype=torch.float64)
    X[3, 1] = float("inf")
    with pytest.raises(ValueError, match="Sample covariances contains Inf or NaN"):
        ops.population_factor(X, w)


def test_resume_with_other_chain_count_is_refused(tmp_path):
    """load_stage: a stage written for 40 chains cannot seed a 30-chain sampler (this used
    to reset the step sizes and the Philox step counter silently)"""
    from beat_amd.sampler import SMC, smc_sample
    from beat_amd.sampler.hosttarget import HostTarget
    from beat_amd.sampler.smc import load_stage
    f, n = _two_gaussians()
    step = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=40, tune_interval=5, random_seed=1)
    smc_sample(6, step, homepath=str(tmp_path), max_stages=1)
    other = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=30, tune_interval=5, random_seed=1)
    with pytest.raises(ValueError, match="resume with the same n_chains"):
        load_stage(other, str(tmp_path), 0)
