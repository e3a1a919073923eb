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


def running_window_rms_batch(data, window_size):
    """utility.running_window_rms(mode="same") of every row of a torch tensor (nd, n), on its device:
    sqrt of the running mean of squares, the window centred like numpy.convolve(..., "same")"""
    import torch
    nd, n = data.shape
    w = int(window_size)
    pad = torch.zeros((nd, w - 1), dtype=data.dtype, device=data.device)
    d2 = torch.cat([pad, data * data, pad], 1)
    cs = torch.cat([torch.zeros((nd, 1), dtype=data.dtype, device=data.device), torch.cumsum(d2, 1)], 1)
    full = (cs[:, w:] - cs[:, :-w]) / float(w)          # "full" convolution with ones(w) / w: n + w - 1 values
    o = (w - 1) // 2
    return torch.sqrt(torch.clamp(full[:, o:o + n], min=0.0))


class NoiseCovarianceUpdate(object):
    """The ``update`` argument of ``smc_sample``: at the end of every stage the reference calls
    ``update.update_weights(map_pt)`` and re-evaluates the population (sampler/smc.py:492-503).
    For the seismic composite with ``noise_estimator.structure == "non-toeplitz"`` that is
    (models/seismic.py:1509-1534, covariance.py:307-325, 397-427, heart.py:211-253):

        synthetics at the MAP point -> residuals per dataset -> window = n // 5 ->
        stds = running_window_rms -> coeffs = autocovariance(res / stds) ->
        C = toeplitz(coeffs) * stds stds^T -> ensure_cov_psd -> chol_inverse, log_pdet -> weights

    Everything up to the weights stays on the device (``beatamd_ffi_synthetics_batch``,
    ``_autocovariance_batch``, ``_scaled_toeplitz_batch``, ``_chol_inverse_batch_flags``).  The
    positive-definiteness check IS the device factorisation: only matrices it flags go to the host
    for the reference's eigenvalue repair (utility.repair_covariance, utility.py:1113-1138) and are
    factored again.  ``last_ms`` / ``n_repaired`` report the last update."""

    def __init__(self, logp_func):
        self.f = logp_func
        self.last_ms, self.n_repaired, self.n_updates = 0.0, 0, 0

    def data_covariances(self, q_map, wavemap_index=0):
        """-> (covariances [T, n, n] torch-cuda, residuals [T, n]) at the point q_map"""
        import torch
        f = self.f
        dev = torch.device("cuda", f.ctx.device)
        q = torch.as_tensor(np.ascontiguousarray(q_map, dtype=np.float64).reshape(1, -1)).to(dev)
        res = f.synthetics(q, wavemap_index, residuals=True)[0]            # seismic.py:1332
        wm = f.problem.wavemaps[wavemap_index]
        if getattr(wm, "is_prewhitened", False):
            # library rows and data of a pre-whitened model hold W_old G and W_old d: the model's residuals are
            # W_old (d - s).  The reference estimates the noise on d - s (covariance.py:307-325,
            # seismic.py:1509-1534): take W_old off again, r_t = inv(W_old,t) . res_t
            res = self._unwhiten(res, wm._whitened_with, dev)
        n = int(res.shape[1])
        window = n // 5
        if window == 0:
            raise ValueError("Length of trace too short! Please widen taper in time domain or frequency bands "
                             "in spectral domain.")                     # covariance.py:317-321
        stds = running_window_rms_batch(res, window)
        coeffs = f.ctx.autocovariance_batch((res / stds).contiguous())
        return f.ctx.scaled_toeplitz_batch(coeffs, stds.contiguous()), res

    def _unwhiten(self, res, w_old, dev):
        """rows r_t = inv(W_t) res_t for upper-triangular W [T, n, n] (host or device): one back substitution per
        dataset on the device (``beatamd_unwhiten_traces``; the first version formed the 64 inverses for this)"""
        import torch
        out = res.clone().contiguous()
        wo = w_old.to(dev) if torch.is_tensor(w_old) else torch.from_numpy(np.ascontiguousarray(w_old)).to(dev)
        self.f.ctx.unwhiten_traces(wo.contiguous(), out)
        return out

    def update_weights(self, q_map):
        import time
        import torch
        from .utility import repair_covariance
        f = self.f
        f.ctx.synchronize()
        t0 = time.perf_counter()
        self.n_repaired = 0
        for i, wm in enumerate(f.problem.wavemaps):
            covs, _ = self.data_covariances(q_map, i)
            W, logdet, bad = f.ctx.chol_inverse_batch_flags(covs)
            bad_idx = torch.nonzero(bad).ravel().tolist()
            for t in bad_idx:                                           # utility.ensure_cov_psd
                fixed = repair_covariance(covs[t].cpu().numpy())
                Wt, lt = f.ctx.chol_inverse_batch(torch.from_numpy(fixed[None]).to(covs.device))
                W[t], logdet[t] = Wt[0], lt[0]
            self.n_repaired += len(bad_idx)
            f.update_weights(i, W, logdet)
        f.ctx.synchronize()
        self.last_ms = (time.perf_counter() - t0) * 1e3
        self.n_updates += 1
