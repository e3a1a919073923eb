"""
Proposal set-up shared by the samplers.

The reference draws ``n_steps`` proposal rows per chain from a numpy distribution object at the
start of every stage (beat/sampler/metropolis.py:289-292 with the classes of
beat/sampler/base.py:74-224).  Here a proposal is a *factor* ``F`` (K x nparams) with
``F^T F = covariance``; one step of all chains draws ``delta = z . F`` on the device
(``beatamd_proposal_draw``: Philox normals + FP64 MFMA GEMM).  Two ways to obtain the factor:

  * from the weighted population itself (``beatamd_smc_population_factor``), K = number of chains
    -- no covariance matrix is formed or factored (SMC stage transition);
  * from a given covariance matrix (PT, user input): one host Cholesky at set-up time,
    K = nparams; a matrix that is not positive definite is repaired like
    ``utility.repair_covariance`` (utility.py:1113-1138) first.
"""
import numpy as np

from ..utility import ensure_cov_psd
from .ops import step_tune  # noqa: F401  (re-export: pymc's tune table)

multivariate_proposals = ("MultivariateNormal", "MultivariateCauchy")
# per-parameter families (base.py:129-155): kernel kinds of beatamd_proposal_draw_univariate
univariate_proposals = {"Normal": 0, "Cauchy": 1, "Laplace": 2, "Poisson": 3}
available_proposals = multivariate_proposals + tuple(univariate_proposals)


def proposal_df(proposal_name):
    """degrees of freedom of the multivariate t the proposal is drawn from (0 = normal);
    MultivariateCauchy = t with one degree of freedom (base.py:177-186); None for the
    per-parameter families (Normal, Cauchy, Laplace, Poisson: base.py:129-155)."""
    if proposal_name in univariate_proposals:
        return None
    if proposal_name not in multivariate_proposals:
        raise NotImplementedError("GPU samplers draw %s proposals, not %s"
                                  % (" / ".join(available_proposals), proposal_name))
    return 1 if proposal_name == "MultivariateCauchy" else 0


def covariance_factor(cov):
    """(nparams, nparams) factor F with F^T F = cov (set-up time, host LAPACK)"""
    cov = np.atleast_2d(np.asarray(cov, dtype=np.float64))
    if not np.isfinite(cov).all():
        raise ValueError("Sample covariances contains Inf or NaN! Please try reducing the"
                         " upper and lower bounds of hyper parameters!")
    return np.ascontiguousarray(np.linalg.cholesky(ensure_cov_psd(cov)).T)
