"""Known-answer tests restated from the reference's own test-suite, applied to the
CPU oracle.  CPU only.  Each test cites the reference test it follows."""
import numpy as np
import scipy.stats

from oracle import oracle as orc


def test_fastsweep_kat():
    """reference test/test_fastsweep.py:20-133: 6 (dip) x 4 (strike) grid, nucleation
    strike idx 2 / dip idx 3, velocity 1.0 in the left half and 3.5 in the right half,
    patch 10 km; closed-form checks on the result."""
    n_dip, n_strike, psz = 6, 4, 10.0
    velo = np.concatenate((np.ones((n_dip, 2)), np.ones((n_dip, 2)) * 3.5), axis=1)
    slow = (1.0 / velo).ravel()
    # Sweeper.perform argument order (pytensorf.py:473-482): (nuc_dip, nuc_strike, n_dip, n_strike)
    t = orc.fast_sweep(slow, psz, 3, 2, n_dip, n_strike).reshape(n_dip, n_strike)
    assert t[3, 2] == 0.0
    # along the nucleation column (strike idx 2, fast half): one-sided updates s*h
    np.testing.assert_allclose(t[:, 2], np.abs(np.arange(n_dip) - 3) * psz / 3.5, atol=1e-12)
    # neighbour in the fast half along strike
    np.testing.assert_allclose(t[3, 3], psz / 3.5, atol=1e-12)
    assert (t >= 0).all() and np.isfinite(t).all()


def _toy(n_datasets=2, n_samples=10, seed=1):
    rng = np.random.default_rng(seed)
    ydata = [rng.standard_normal(n_samples) for _ in range(n_datasets)]
    syn = [y + 0.03 * rng.standard_normal(n_samples) for y in ydata]
    return ydata, syn


def test_mvn_chol_vs_scipy_identity():
    """reference test/test_models.py:149-222: 2 datasets x 10 samples, C = 0.001*I,
    hyperparameter 0.0 -> logp == scipy multivariate_normal.logpdf (atol 1e-6)"""
    ydata, syn = _toy()
    C = 0.001 * np.eye(10)
    W = orc.cov_chol_inverse(C)
    slog = orc.cov_log_pdet(C)
    for y, s in zip(ydata, syn):
        ref = scipy.stats.multivariate_normal.logpdf(y, mean=s, cov=C)
        np.testing.assert_allclose(orc.mvn_chol_logp(W, y - s, slog, 0.0), ref, rtol=0, atol=1e-6)
        # scalar-weight fast path: chol_inverse of sigma^2 I is I / sigma
        np.testing.assert_allclose(orc.mvn_chol_logp(1.0 / np.sqrt(0.001), y - s, slog, 0.0), ref,
                                   rtol=0, atol=1e-6)


def test_mvn_chol_vs_scipy_dense_and_hyper():
    """same check with a dense (Toeplitz) covariance and a non-zero hyperparameter:
    scaling the covariance by exp(2h) (distributions.py:129-136)"""
    ydata, syn = _toy(n_samples=48, seed=5)
    C = 0.5 * orc.exponential_data_covariance(48, 0.5, 2.0)
    W = orc.cov_chol_inverse(C)
    slog = orc.cov_log_pdet(C)
    for hp in (0.0, 0.7, -1.3):
        for y, s in zip(ydata, syn):
            ref = scipy.stats.multivariate_normal.logpdf(y, mean=s, cov=C * np.exp(2 * hp))
            np.testing.assert_allclose(orc.mvn_chol_logp(W, y - s, slog, hp), ref, rtol=1e-10,
                                       atol=1e-6)


def test_covariance_chol_inverse_kat():
    """reference test/test_covariance.py:71-112: W.T W == inv(C), atol 1e-6"""
    np.random.seed(10)
    n = 10
    a = np.random.rand(n ** 2).reshape(n, n)
    C_d = a.T.dot(a) + np.eye(n) * 0.3
    W = orc.cov_chol_inverse(C_d)
    np.testing.assert_allclose(W.T.dot(W), np.linalg.inv(C_d), rtol=0, atol=1e-6)
    assert np.allclose(W, np.triu(W))  # upper right factor


def test_log_pdet_vs_scipy_psd():
    """reference test/test_models.py:163-166"""
    C = 0.001 * np.eye(10)
    psd = scipy.stats._multivariate._PSD(C)
    np.testing.assert_allclose(orc.cov_log_pdet(C), psd.log_pdet, rtol=0, atol=1e-6)


def test_stack_closed_form():
    """reference test/test_ffi.py:22-89 library recipe: T=30, P=40, D=11, S=31, N=10,
    every trace of target i is arange(N)*i  ->  stack_all[t, n] = t * n * sum(slips)
    (nn) and the same for multilinear because the blend weights sum to slip."""
    T, P, D, S, N = 30, 40, 11, 31, 10
    G = np.empty((T, P, D, S, N))
    G[:] = (np.arange(N)[None, :] * np.arange(T)[:, None])[:, None, None, None, :]
    rng = np.random.default_rng(0)
    dur = rng.uniform(5.0, 10.0, P)
    st = rng.uniform(0.0, 15.0, (T, P))
    sl = rng.random(P)
    expect = np.arange(T)[:, None] * np.arange(N)[None, :] * sl.sum()
    for interp in ("nearest_neighbor", "multilinear"):
        out = orc.stack_all(G, dur, st, sl, 5.0, 0.5, 0.0, 0.5, interp)
        np.testing.assert_allclose(out, expect, rtol=1e-12, atol=1e-10)


def test_stack_out_of_bounds_raises():
    G = np.zeros((2, 3, 2, 2, 4))
    try:
        orc.stack_all(G, np.full(3, 5.0), np.zeros((2, 3)), np.ones(3), 0.5, 0.5, 0.0, 0.5)
    except IndexError:
        return
    raise AssertionError("expected IndexError like numpy fancy indexing")


def test_metrop_and_pt_decisions():
    """metropolis.py:355-358 / pt.py:429-457 decision rules"""
    assert orc.metrop_accept(0.5, -10.0, -12.0, np.log(0.5))       # mr = +1
    assert not orc.metrop_accept(0.5, -14.0, -12.0, np.log(0.5))   # mr = -1 < log .5? no: -0.69 > -1
    assert not orc.metrop_accept(1.0, np.nan, -12.0, -100.0)       # non-finite never accepted
    assert orc.pt_swap_accept(1.0, 0.5, -20.0, -10.0, np.log(0.9))  # alpha = +5
    assert not orc.pt_swap_accept(1.0, 0.5, -10.0, -20.0, np.log(0.9))


def test_whitening_operator_of_the_exponential_structure_is_bidiagonal():
    """the premise of the banded evaluation (beatamd_weights_band): the reference's "exponential" noise
    structure (beat/covariance.py:24-51: C_ij = exp(-|i-j| dt / t0)) is a Markov kernel -- inv(C) is tridiagonal, so
    W = cholesky(inv(C)).T (heart.py:211-237) is BIDIAGONAL; what numpy's inv + cholesky leave outside the band is rounding
    residue far below the library's threshold 2^-40 of the largest entry.  Its closed form (AR(1) innovations):
    W[i,i] = 1/sqrt(1-a^2), W[i,i+1] = -a/sqrt(1-a^2) for i < n-1, W[n-1,n-1] = 1, a = exp(-dt/t0)"""
    from oracle import oracle as orc
    n, dt, t0 = 700, 0.5, 2.0
    C = orc.exponential_data_covariance(n, dt, t0)
    W = orc.cov_chol_inverse(3.7 * C)
    big = np.abs(W).max()
    assert np.abs(np.tril(W, -1)).max() == 0.0
    assert np.abs(np.triu(W, 2)).max() < 2.0 ** -40 * big and np.abs(np.triu(W, 2)).max() < 1e-13 * big
    a = np.exp(-dt / t0)
    d = np.full(n, 1.0 / np.sqrt(1.0 - a * a))
    d[-1] = 1.0
    np.testing.assert_allclose(np.diag(W) * np.sqrt(3.7), d, rtol=1e-11)
    np.testing.assert_allclose(np.diag(W, 1) * np.sqrt(3.7), np.full(n - 1, -a / np.sqrt(1.0 - a * a)), rtol=1e-11)
    # a covariance with a dense term added (a model-prediction covariance) has no band: the dense kernel's case
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, 40))
    Wd = orc.cov_chol_inverse(C + 0.05 * A @ A.T / 40.0)
    assert np.abs(np.triu(Wd, 17)).max() > 1e-6 * np.abs(Wd).max()
