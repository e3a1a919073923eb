"""Host helpers with the reference's names (beat/utility.py)."""
import numpy as np


def positions2idxs(positions, cell_size, min_pos=0.0, backend=np, dtype="int16"):
    """utility.py:1542-1558 (round-half-even, int16)"""
    return backend.round((positions - min_pos - (cell_size / 2.0)) / cell_size).astype(dtype)


def ensure_cov_psd(cov):
    """utility.py:1034-1056"""
    try:
        np.linalg.cholesky(cov)
        return cov
    except np.linalg.LinAlgError:
        return repair_covariance(cov)


def repair_covariance(x, epsilon=np.finfo(np.float64).eps):
    """utility.py:1113-1138: clip the eigenvalues at epsilon and transform back"""
    eigval, eigvec = np.linalg.eigh(x)
    val = np.maximum(eigval, epsilon)
    return eigvec.dot(np.diag(val)).dot(eigvec.T)


def running_window_rms(data, window_size, mode="valid"):
    """utility.py:1141-1161"""
    data2 = np.power(data, 2)
    window = np.ones(window_size) / float(window_size)
    return np.sqrt(np.convolve(data2, window, mode))
