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


def test_smc_methods_match_reference_golden():
    """calc_beta / resample / weighted covariance against arrays captured from the
    reference's SMC methods (oracle/gen_golden.py)"""
    from beat_amd.sampler import SMC
    from beat_amd.sampler.hosttarget import HostTarget
    g = load_golden("smc")
    for k in range(int(g["ncase"])):
        lk = g["c%d_lk" % k]
        step = SMC(HostTarget(lambda X: np.zeros(len(X)), 6), -np.ones(6), np.ones(6),
                   n_chains=lk.size)
        step.likelihoods = lk
        step.beta = float(g["c%d_beta_in" % k])
        b, ob, w = step.calc_beta()
        assert b == float(g["c%d_beta" % k]) and ob == step.beta
        np.testing.assert_array_equal(w, g["c%d_w" % k])
        step.weights = w
        aux = g["c%d_aux" % k]
        step.rng = type("R", (), {"rand": staticmethod(lambda n, a=aux: a)})()
        assert np.array_equal(step.resample(), g["c%d_idx" % k])
        step.array_population = g["c%d_pop" % k]
        np.testing.assert_allclose(step.calc_covariance(), g["c%d_cov" % k], rtol=1e-12, atol=1e-15)


def test_tune_tables():
    from beat_amd.sampler import pt, smc
    from beat_amd.sampler.base import step_tune
    g = load_golden("smc")
    for a, p, s in zip(g["tune_acc"], g["pt_tune"], g["smc_tune"]):
        assert pt.tune(1.2, a) == p and smc.tune(a) == s
    np.testing.assert_allclose(step_tune(np.ones(6), [0.0005, 0.03, 0.1, 0.3, 0.8, 0.99]),
                               [0.1, 0.5, 0.9, 1.0, 2.0, 10.0])


def test_metropolis_astep_semantics():
    """metropolis.py:276-422 decision sequence on a 1-chain function"""
    from beat_amd.models import prior_logp_func
    from beat_amd.sampler import Metropolis
    from oracle import oracle as orc
    lo, up = -np.ones(3), np.ones(3)
    calls = []

    def logp(q):
        calls.append(q.copy())
        return [np.array(-50.0 * np.sum(q ** 2))]

    m = Metropolis(logp, prior_logp_func(lo, up), 3, n_chains=1, tune_interval=5,
                   proposal_scale=np.eye(3) * 0.01)
    q0 = np.zeros(3)
    q, l = m.astep(q0)  # stage 0: evaluate, no move
    assert np.array_equal(q, q0) and len(calls) == 1
    m.stage, m.n_steps, m.beta = 1, 20, 0.5
    np.random.seed(4)
    n_acc = 0
    for i in range(20):
        ncalls = len(calls)
        u_state = np.random.get_state()
        q_new, l_new = m.astep(q)
        moved = not np.array_equal(q_new, q)
        if moved:
            # accepted proposals satisfy the oracle's rule for some u: mr finite
            assert orc.metrop_accept(0.5, l_new[-1], l[-1], -np.inf)
        n_acc += moved
        q, l = q_new, l_new
    assert m.stage_sample == 0 and m.cumulative_samples == 20 and 0 < n_acc <= 20
    # outside the prior box: no forward evaluation, chain stays (metropolis.py:341-343,383-385)
    m.proposal_samples_array = np.full((20, 3), 100.0)
    m.stage_sample = 1
    ncalls = len(calls)
    q2, l2 = m.astep(q)
    assert np.array_equal(q2, q) and len(calls) == ncalls
    # NaN likelihood at stage 0 raises (metropolis.py:279-284)
    bad = Metropolis(lambda q: [np.array(np.nan)], prior_logp_func(lo, up), 3)
    try:
        bad.astep(q0)
        raise AssertionError("expected ValueError")
    except ValueError:
        pass


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
    # resume after stage 1: same remaining stage count (deterministic shared RNG differs, so only
    # structure is compared)
    step2 = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=64, tune_interval=10,
                random_seed=4)
    pop2, lp2, betas2 = smc_sample(20, step2, homepath=str(tmp_path), layout=None, resume_stage=1)
    assert betas2[0] == betas[1] and betas2[-1] == 1.0 and pop2.shape == pop.shape
    assert nstage >= 2


def test_device_proposals_normal_and_cauchy_statistics():
    """DeviceMvNormalProposal on the torch CPU device (the same code draws on the GPU): rows have the
    requested covariance (MultivariateNormal), resp. are multivariate t with one degree of
    freedom (MultivariateCauchy, base.py:35-71,163-186: z / sqrt(chi2(1)))"""
    import torch
    from beat_amd.sampler.base import DeviceMvNormalProposal
    rng = np.random.default_rng(3)
    A = rng.standard_normal((4, 4))
    cov = A @ A.T + 0.5 * np.eye(4)
    pn = DeviceMvNormalProposal(cov, torch.device("cpu"), seed=5)
    x = pn(200000).numpy()
    np.testing.assert_allclose(np.cov(x.T), cov, rtol=0.03, atol=0.03)
    pc = DeviceMvNormalProposal(cov, torch.device("cpu"), seed=6, df=1)
    y = pc(200000).numpy()
    # marginals of a multivariate Cauchy are Cauchy with scale sqrt(cov_ii): median |y_i| = scale
    np.testing.assert_allclose(np.median(np.abs(y), axis=0), np.sqrt(np.diag(cov)), rtol=0.03)
    # heavy tails: P(|y| > 10 scale) = 1 - 2/pi atan(10) = 0.0635
    frac = (np.abs(y[:, 0]) > 10 * np.sqrt(cov[0, 0])).mean()
    assert abs(frac - 0.0635) < 0.004
    # a singular covariance (population smaller than the parameter count) is repaired, not rejected
    B = rng.standard_normal((6, 2))
    ps = DeviceMvNormalProposal(B @ B.T, torch.device("cpu"), seed=1)
    z = ps(50000).numpy()
    np.testing.assert_allclose(np.cov(z.T), B @ B.T, rtol=0.05, atol=0.05)
    with pytest.raises(ValueError):
        DeviceMvNormalProposal(cov, torch.device("cpu"), df=0.5)(3)


def test_population_proposal_equals_weighted_covariance():
    """DeviceMvNormalProposal.from_population draws with np.cov(X, aweights=w) (SMC.calc_covariance)
    without factoring it, also when the population is smaller than the parameter count"""
    import torch
    from beat_amd.sampler.base import DeviceMvNormalProposal
    rng = np.random.default_rng(11)
    for n, d in ((40, 6), (5, 9)):
        X = rng.standard_normal((n, d)) * rng.uniform(0.5, 3.0, d) + rng.standard_normal(d)
        w = rng.random(n)
        w /= w.sum()
        cov = np.cov(X, aweights=w, bias=False, rowvar=0)
        prop = DeviceMvNormalProposal.from_population(X, w, torch.device("cpu"), seed=2)
        np.testing.assert_allclose(prop.LT.numpy().T @ prop.LT.numpy(), cov, rtol=1e-12, atol=1e-12)
        rows = prop(300000).numpy()
        assert rows.shape == (300000, d)
        np.testing.assert_allclose(np.cov(rows.T), cov, rtol=0.04, atol=0.04 * np.abs(cov).max())
