"""
Proposal distributions and tuning tables of the samplers -- host-side counterparts of
beat/sampler/base.py:35-224 (same class names and call conventions), plus the batched
device proposal used by the GPU samplers.
"""
import numpy as np
from numpy.random import (normal, poisson, randint, standard_cauchy,  # noqa: F401
                          standard_exponential)


def multivariate_t_rvs(mean, cov, df=np.inf, size=1):
    """base.py:35-71"""
    m = np.asarray(mean)
    d = len(mean)
    x = 1.0 if df == np.inf else np.random.chisquare(df, size) / df
    z = np.random.multivariate_normal(np.zeros(d), cov, (size,))
    return m + z / np.sqrt(x)[:, None]


class Proposal(object):
    def __init__(self, scale):
        self.scale = np.atleast_1d(scale)


class DiscreteBoundedUniformProposal(Proposal):
    """base.py:87-125"""

    def __init__(self, lower=0, upper=10, scale=1):
        self.lower, self.upper = lower, upper
        super(DiscreteBoundedUniformProposal, self).__init__(scale)

    def __call__(self, size=1):
        return (randint(low=self.upper - self.lower, size=size) + self.lower) * self.scale


class NormalProposal(Proposal):
    def __call__(self, num_draws=None):
        size = self.scale.shape
        if num_draws:
            size += (num_draws,)
        return normal(scale=self.scale[0], size=size).T


class CauchyProposal(Proposal):
    def __call__(self, num_draws=None):
        size = self.scale.shape
        if num_draws:
            size += (num_draws,)
        return standard_cauchy(size=size).T * self.scale


class LaplaceProposal(Proposal):
    def __call__(self, num_draws=None):
        size = self.scale.shape
        if num_draws:
            size += (num_draws,)
        return (standard_exponential(size=size) - standard_exponential(size=size)).T * self.scale


class PoissonProposal(Proposal):
    def __call__(self, num_draws=None):
        size = self.scale.shape
        if num_draws:
            size += (num_draws,)
        return poisson(lam=self.scale, size=size).T - self.scale


class MultivariateNormalProposal(Proposal):
    def __call__(self, num_draws=None):
        return np.random.multivariate_normal(mean=np.zeros(self.scale.shape[0]), cov=self.scale,
                                             size=num_draws)


class MultivariateCauchyProposal(Proposal):
    def __call__(self, num_draws=None):
        return multivariate_t_rvs(mean=np.zeros(self.scale.shape[0]), cov=self.scale, df=1,
                                  size=num_draws)


proposal_distributions = {
    "Cauchy": CauchyProposal,
    "Poisson": PoissonProposal,
    "Normal": NormalProposal,
    "Laplace": LaplaceProposal,
    "MultivariateNormal": MultivariateNormalProposal,
    "MultivariateCauchy": MultivariateCauchyProposal,
    "DiscreteBoundedUniform": DiscreteBoundedUniformProposal,
}
multivariate_proposals = ["MultivariateCauchy", "MultivariateNormal"]


def available_proposals():
    return list(proposal_distributions.keys())


def choose_proposal(proposal_name, **kwargs):
    """base.py:207-224"""
    return proposal_distributions[proposal_name](**kwargs)


def step_tune(scale, acc_rate):
    """pymc.step_methods.metropolis.tune used by Metropolis.astep (metropolis.py:294-306).
    pymc is not in the reference tree; table restated from its documentation:
        <0.001 x0.1 | <0.05 x0.5 | <0.2 x0.9 | >0.95 x10 | >0.75 x2 | >0.5 x1.1
    Vectorised over chains."""
    scale = np.asarray(scale, dtype=np.float64)
    acc = np.asarray(acc_rate, dtype=np.float64)
    f = np.ones_like(acc)
    f = np.where(acc > 0.5, 1.1, f)
    f = np.where(acc > 0.75, 2.0, f)
    f = np.where(acc > 0.95, 10.0, f)
    f = np.where(acc < 0.2, 0.9, f)
    f = np.where(acc < 0.05, 0.5, f)
    f = np.where(acc < 0.001, 0.1, f)
    return scale * f


def metrop_select(mr, q, q0):
    """pymc metrop_select semantics used at metropolis.py:358 (not in tree):
    accept iff isfinite(mr) and log(uniform) < mr."""
    if np.isfinite(mr) and np.log(np.random.uniform()) < mr:
        return q, True
    return q0, False


class DeviceMvNormalProposal(object):
    """MultivariateNormal (df = inf) / MultivariateCauchy (df = 1) proposal rows generated on the GPU (replaces the per-chain
    ``proposal_dist(n_steps)`` host draws of metropolis.py:289-292): rows = z @ chol(cov).T.
    torch is plumbing here (RNG + one library GEMM per step)."""

    def __init__(self, cov, device, seed=0, df=np.inf):
        import torch
        self.df = float(df)   # inf: MultivariateNormal; 1: MultivariateCauchy (base.py:163-186)
        covd = torch.as_tensor(np.atleast_2d(np.asarray(cov, dtype=np.float64))).to(device)
        # factor on the device; a population smaller than the parameter count gives a singular
        # sample covariance: repair it like utility.repair_covariance (eigenvalues clipped at
        # machine epsilon, utility.py:1113-1138) and take L = V sqrt(lambda) -- one eigh instead
        # of the reference's Cholesky attempt + eigh repair + SVD inside multivariate_normal
        L, info = torch.linalg.cholesky_ex(covd)
        if int(info.item()) != 0 or not bool(torch.isfinite(L).all()):
            w, v = torch.linalg.eigh(covd)
            L = v * torch.sqrt(torch.clamp(w, min=float(np.finfo(np.float64).eps)))
        self.LT = L.T.contiguous()
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))
        self.device = device

    @classmethod
    def from_population(cls, population, weights, device, seed=0, df=np.inf):
        """Proposal with the weighted sample covariance of ``population`` (n, nparams) --
        ``np.cov(population, aweights=weights, bias=False, rowvar=0)`` as in SMC.calc_covariance
        (smc.py:167-186) -- WITHOUT forming or factoring it: with
        X_c = sqrt(w / (1 - sum w^2)) (x - weighted mean), rows = z @ X_c (z of n standard normals)
        have exactly that covariance.  For populations smaller than the parameter count the
        sample covariance is singular and the factorisation route ends in an eigendecomposition
        per stage (tens of ms for 1200 parameters); this route is one GEMM per step either way."""
        import torch
        self = cls.__new__(cls)
        self.df = float(df)
        X = torch.as_tensor(np.asarray(population, dtype=np.float64)).to(device)
        w = torch.as_tensor(np.asarray(weights, dtype=np.float64).ravel()).to(device)
        w = w / w.sum()
        mean = (w[:, None] * X).sum(0)
        fact = torch.sqrt(w / (1.0 - (w * w).sum()))
        self.LT = (fact[:, None] * (X - mean)).contiguous()      # (n, nparams)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))
        self.device = device
        return self

    def __call__(self, n_chains):
        import torch
        z = torch.randn((n_chains, self.LT.shape[0]), generator=self.gen, device=self.device,
                        dtype=torch.float64)
        rows = z @ self.LT
        if np.isfinite(self.df):
            # multivariate_t_rvs (base.py:35-71): z / sqrt(chi2(df) / df), one draw per row
            k = int(self.df)
            if k != self.df or k < 1:
                raise ValueError("degrees of freedom must be a positive integer")
            g = torch.randn((n_chains, k), generator=self.gen, device=self.device, dtype=torch.float64)
            x = (g * g).sum(1) / self.df
            rows = rows / torch.sqrt(x)[:, None]
        return rows

    def log_uniform(self, n_chains):
        import torch
        return torch.log(torch.rand((n_chains,), generator=self.gen, device=self.device,
                                    dtype=torch.float64))
