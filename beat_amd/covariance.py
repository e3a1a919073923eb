"""
Noise-covariance structures and estimators -- counterpart of beat/covariance.py (the parts
that feed the sampler's likelihood weights).  The O(n^2) estimators run on the GPU.
"""
import numpy as np

from .engine import get_context
from .utility import ensure_cov_psd, running_window_rms  # noqa: F401


def exponential_data_covariance(n, dt, tzero):
    """covariance.py:24-51: exp(-|ti - tj| / T0), Toeplitz"""
    i = np.arange(n)
    return np.exp(-np.abs(i[:, np.newaxis] - i[np.newaxis, :]) * (dt / tzero))


def identity_data_covariance(n, dt=None, tzero=None):
    """covariance.py:54-67"""
    return np.eye(n, dtype="float64")


def ones_data_covariance(n, dt=None, tzero=None):
    """covariance.py:70-83"""
    return np.ones((n, n), dtype="float64")


NoiseStructureCatalog = {
    "variance": identity_data_covariance,
    "exponential": exponential_data_covariance,
    "import": ones_data_covariance,
    "non-toeplitz": ones_data_covariance,
}


def available_noise_structures():
    return list(NoiseStructureCatalog.keys())


def autocovariance(data):
    """covariance.py:716-736 (bitwise the reference's double loop, evaluated on the GPU)"""
    d = np.ascontiguousarray(data, dtype=np.float64).reshape(1, -1)
    return get_context().autocovariance_batch(d)[0]


def autocovariance_batch(data):
    """data (nd, n) -> (nd, n); numpy or torch-cuda"""
    return get_context().autocovariance_batch(data)


def toeplitz_covariance(data, window_size):
    """covariance.py:739-751 -> (toeplitz matrix, stds)"""
    from scipy.linalg import toeplitz
    stds = running_window_rms(data, window_size=window_size, mode="same")
    coeffs = autocovariance(data / stds)
    return toeplitz(coeffs), stds


def non_toeplitz_covariance(data, window_size):
    """covariance.py:754-771: scaled non-Toeplitz covariance for non-stationary errors"""
    return non_toeplitz_covariance_batch(np.asarray(data, dtype=np.float64).reshape(1, -1),
                                         window_size)[0]


def non_toeplitz_covariance_batch(data, window_size):
    """All datasets of a wavemap at once: data (nd, n) -> (nd, n, n).  This is what
    SeismicNoiseAnalyser.do_non_toeplitz (covariance.py:333-395) computes per trace at every
    stage when update_covariances is on."""
    data = np.ascontiguousarray(data, dtype=np.float64)
    stds = np.stack([running_window_rms(d, window_size=window_size, mode="same") for d in data])
    ctx = get_context()
    coeffs = ctx.autocovariance_batch(np.ascontiguousarray(data / stds))
    return ctx.scaled_toeplitz_batch(coeffs, stds)


def get_data_covariances(structure, scalings):
    """covariance.py:413-427: cov_d = ensure_cov_psd(scaling * structure) per dataset"""
    return [ensure_cov_psd(s * structure) for s in scalings]
