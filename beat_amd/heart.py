"""
From a dataset's covariance terms to the sampler's likelihood weights.

The log-likelihood kernels consume, per dataset, the whitening matrix ``W`` (upper triangular,
``W^T W = inv(C)``) and ``log|C|`` (``multivariate_normal_chol``,
beat/models/distributions.py:119-138).  In the reference both come from ``heart.Covariance``
(beat/heart.py:104-263: ``chol_inverse`` :211-237, ``log_pdet`` :239-245, the shared scalar
``slog_pdet`` :247-253).  This module builds the same two quantities once per stage:

  ``whitening(C)``          one covariance      -> (W, log_pdet)          host LAPACK
  ``chol_inverse_batch``    a stack (nd, n, n)  -> (W, log_pdet) arrays   on the GPU
  ``Covariance``            the container the composites hand around: the three additive terms
                            (data, pred_g, pred_v) and the derived weights under the
                            reference's attribute names, so code written against
                            ``dataset.covariance.chol_inverse`` / ``.slog_pdet`` keeps working.
"""
import numpy as np

TERMS = ("data", "pred_g", "pred_v")


def log_determinant(A, inverse=False):
    """log|A| of a positive definite matrix through its Cholesky factor (heart.py:65-89);
    ``inverse`` gives log|inv(A)| = -log|A| the way the reference forms it"""
    half = np.log(np.diag(_lower_factor(A, inverse))).sum()
    return 2.0 * half


def _lower_factor(A, inverted=False):
    from scipy.linalg import cholesky
    L = cholesky(np.asarray(A, dtype=np.float64), lower=True)
    return np.linalg.inv(L) if inverted else L


def whitening(C):
    """(W, log_pdet) of one total covariance: W = cholesky(inv(C)).T, upper triangular with
    W^T W = inv(C) (heart.py:231-233); when inv(C) is numerically not positive definite the R
    factor of QR(inv(chol(C))) is used instead -- same W^T W (heart.py:234-237).
    log_pdet = 2 sum log diag chol(C) (heart.py:239-245)."""
    C = np.asarray(C, dtype=np.float64)
    if not C.any():
        raise ValueError("No covariances given!")
    L = _lower_factor(C)
    try:
        W = np.linalg.cholesky(np.linalg.inv(C)).T
    except np.linalg.LinAlgError:
        W = np.linalg.qr(np.linalg.inv(L))[1]
    return np.ascontiguousarray(W, dtype=np.float64), np.float64(2.0 * np.log(np.diag(L)).sum())


class _Scalar(object):
    """get_value / set_value holder standing in for the reference's shared scalar"""

    def __init__(self, name, value=0.0):
        self.name, self._v = name, value

    def get_value(self, borrow=False):
        return self._v

    def set_value(self, value, borrow=False):
        self._v = value

    def __float__(self):
        return float(self._v)


class Covariance(object):
    """Additive covariance terms of one dataset: ``data`` (observation noise), ``pred_g``
    (fault-geometry prediction error), ``pred_v`` (velocity-model prediction error); any of the
    latter two may be missing (heart.py:104-156).  Derived quantities are computed on access from
    the current terms."""

    def __init__(self, data=None, pred_g=None, pred_v=None):
        self._terms = dict(data=data, pred_g=pred_g, pred_v=pred_v)
        self.slog_pdet = _Scalar("cov_normalisation")
        if data is not None:
            self.update_slog_pdet()

    # -- the three terms as attributes
    def _get(name):
        return property(lambda self: self._terms[name],
                        lambda self, value: self._terms.__setitem__(name, value))

    data, pred_g, pred_v = _get("data"), _get("pred_g"), _get("pred_v")
    del _get

    def covs_supported(self):
        return [t for t in TERMS[1:]] + [TERMS[0]]

    def _term(self, name):
        """the term as an array of the data covariance's shape (zeros when absent)"""
        if name not in TERMS:
            raise NotImplementedError("Covariance term %s not supported" % name)
        ref = self._terms["data"]
        m = self._terms[name]
        if m is None or (np.size(m) != np.size(ref) and not np.any(m)):
            m = np.zeros_like(ref, dtype="float64")
            self._terms[name] = m
        elif np.size(m) != np.size(ref):
            raise ValueError("%s covariances defined but size inconsistent!" % name)
        return m

    check_matrix_init = _term

    def _sum(self, names):
        return sum(self._term(n) for n in names)

    c_total = property(lambda self: self._sum(TERMS))
    p_total = property(lambda self: self._sum(TERMS[1:]))

    # -- inverses
    @staticmethod
    def _inv(M, what):
        if M is None or not np.any(M):
            raise ValueError(what)
        return np.linalg.inv(M).astype("float64")

    def inverse(self, factor=1.0):
        return self._inv(self.c_total * factor, "No covariances given!")

    inverse_p = property(lambda self: self._inv(self.p_total, "No model covariance defined!"))

    @property
    def inverse_d(self):
        if self._terms["data"] is None:
            raise AttributeError("No data covariance matrix defined!")
        return self._inv(self._terms["data"], "No covariances given!")

    # -- factors and weights
    def chol(self, factor=1.0):
        total = self.c_total * factor
        if not np.any(total):
            raise ValueError("No covariances given!")
        return _lower_factor(total).astype("float64")

    chol_inverse = property(lambda self: whitening(self.c_total)[0])
    log_pdet = property(lambda self: whitening(self.c_total)[1])

    def update_slog_pdet(self):
        self.slog_pdet.set_value(self.log_pdet)


def chol_inverse_batch(covs, device=None):
    """(W (nd,n,n), log_pdet (nd,)) of a stack of covariances at once on the GPU -- what
    ``update_weights`` needs per stage for every dataset of a wavemap (seismic.py:1509-1534).
    Library factorisations (torch.linalg = rocSOLVER) as per-stage set-up; results as numpy."""
    import torch
    dev = device if device is not None else torch.device("cuda", 0)
    C = torch.as_tensor(np.ascontiguousarray(covs), dtype=torch.float64, device=dev)
    L = torch.linalg.cholesky(C)
    log_pdet = 2.0 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(1)
    K = torch.linalg.cholesky(torch.cholesky_inverse(L))   # K K^T = inv(C); W = K^T
    return K.transpose(1, 2).contiguous().cpu().numpy(), log_pdet.cpu().numpy()
