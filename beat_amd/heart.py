"""
``Covariance`` with the reference's interface (beat/heart.py:104-263): the per-dataset
container whose ``chol_inverse`` / ``slog_pdet`` become the sampler's weights.  The
factorisations run once per stage (not per step) with LAPACK on the host exactly as in the
reference; ``chol_inverse_batch`` does a whole wavemap at once on the GPU through
torch.linalg (library LAPACK, plumbing) for large N.
"""
import numpy as np
from scipy import linalg


def log_determinant(A, inverse=False):
    """heart.py:65-89"""
    cholesky = linalg.cholesky(A, lower=True)
    if inverse:
        cholesky = np.linalg.inv(cholesky)
    return np.log(np.diag(cholesky)).sum() * 2.0


class _SharedScalar(object):
    """stand-in for the pytensor shared scalar ``slog_pdet`` (heart.py:131, 247-253)"""

    def __init__(self, value, name):
        self._v, self.name = value, name

    def get_value(self, borrow=False):
        return self._v

    def set_value(self, v, borrow=False):
        self._v = v

    def __float__(self):
        return float(self._v)


class Covariance(object):
    """heart.py:104-263"""

    def __init__(self, data=None, pred_g=None, pred_v=None):
        self.data, self.pred_g, self.pred_v = data, pred_g, pred_v
        self.slog_pdet = _SharedScalar(0.0, "cov_normalisation")
        if data is not None:
            self.update_slog_pdet()

    def covs_supported(self):
        return ["pred_g", "pred_v", "data"]

    def check_matrix_init(self, cov_mat_str=""):
        """heart.py:137-156"""
        if cov_mat_str not in self.covs_supported():
            raise NotImplementedError("Covariance term %s not supported" % cov_mat_str)
        cov_mat = getattr(self, cov_mat_str)
        if cov_mat is None:
            cov_mat = np.zeros_like(self.data, dtype="float64")
        if cov_mat.size != self.data.size:
            if cov_mat.sum() == 0.0:
                cov_mat = np.zeros_like(self.data, dtype="float64")
            else:
                raise ValueError("%s covariances defined but size inconsistent!" % cov_mat_str)
        setattr(self, cov_mat_str, cov_mat)

    @property
    def c_total(self):
        for k in ("data", "pred_g", "pred_v"):
            self.check_matrix_init(k)
        return self.data + self.pred_g + self.pred_v

    @property
    def p_total(self):
        self.check_matrix_init("pred_g")
        self.check_matrix_init("pred_v")
        return self.pred_g + self.pred_v

    def inverse(self, factor=1.0):
        Cx = self.c_total * factor
        if Cx.sum() == 0:
            raise ValueError("No covariances given!")
        return np.linalg.inv(Cx).astype("float64")

    @property
    def inverse_p(self):
        if self.p_total.sum() == 0:
            raise ValueError("No model covariance defined!")
        return np.linalg.inv(self.p_total).astype("float64")

    @property
    def inverse_d(self):
        if self.data is None:
            raise AttributeError("No data covariance matrix defined!")
        return np.linalg.inv(self.data).astype("float64")

    def chol(self, factor=1.0):
        Cx = self.c_total * factor
        if Cx.sum() == 0:
            raise ValueError("No covariances given!")
        return linalg.cholesky(Cx, lower=True).astype("float64")

    @property
    def chol_inverse(self):
        """heart.py:211-237: upper right Cholesky factor of inv(C); QR proxy when inv(C) is
        numerically not positive definite"""
        try:
            return np.linalg.cholesky(self.inverse()).T.astype("float64")
        except np.linalg.LinAlgError:
            inverse_chol = np.linalg.inv(self.chol().T)
            _, chol_ur = np.linalg.qr(inverse_chol.T)
            return chol_ur.astype("float64")

    @property
    def log_pdet(self):
        return np.float64(np.log(np.diag(self.chol())).sum() * 2.0)

    def update_slog_pdet(self):
        self.slog_pdet.set_value(self.log_pdet)


def chol_inverse_batch(covs, device=None):
    """chol_inverse and log_pdet of a stack of covariances (nd, n, n) at once on the GPU
    (torch.linalg = rocSOLVER/hipBLAS; per-stage setup).
    Returns (W (nd,n,n) upper-triangular with W^T W = inv(C), log_pdet (nd,)) as numpy."""
    import torch
    dev = device if device is not None else torch.device("cuda", 0)
    C = torch.as_tensor(np.ascontiguousarray(covs), dtype=torch.float64, device=dev)
    L = torch.linalg.cholesky(C)                      # C = L L^T
    log_pdet = 2.0 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(1)
    # inv(C) = L^-T L^-1 ; its lower Cholesky factor K satisfies K K^T = inv(C); W = K^T
    Cinv = torch.cholesky_inverse(L)
    K = torch.linalg.cholesky(Cinv)
    return K.transpose(1, 2).contiguous().cpu().numpy(), log_pdet.cpu().numpy()
