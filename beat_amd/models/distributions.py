"""
``multivariate_normal_chol`` with the reference's argument list
(beat/models/distributions.py:72-140), evaluated on the GPU for a batch of chains.
"""
import numpy as np

from ..engine import get_context

log_2pi = np.log(2 * np.pi)


def get_hyper_name(dataset):
    """distributions.py:24-25"""
    return "_".join(("h", dataset.typ))


class _Counter(object):
    """beat.utility.Counter: running index per hyper-parameter name (distributions.py:117-126)"""

    def __init__(self):
        self.d = {}

    def __call__(self, name):
        i = self.d.get(name, 0)
        self.d[name] = i + 1
        return i


def multivariate_normal_chol(datasets, weights, hyperparams, residuals, hp_specific=False,
                             sparse=False, ctx=None):
    """
    datasets     list of objects with ``.typ``, ``.samples`` and ``.covariance.slog_pdet``
    weights      list of (M, M) chol_inverse matrices (arrays or objects with get_value())
    hyperparams  dict name -> scalar, or (n,) array when hp_specific; a leading batch axis
                 (C, ...) evaluates C chains at once
    residuals    (n_t, M) or batched (C, n_t, M)
    returns      (n_t,) or (C, n_t) log-likelihoods
    """
    ctx = ctx or get_context()
    res = np.asarray(residuals, dtype=np.float64)
    batched = res.ndim == 3
    if not batched:
        res = res[None]
    C, n_t, M = res.shape
    W = np.stack([np.asarray(w.get_value() if hasattr(w, "get_value") else w, dtype=np.float64)
                  for w in weights])
    slog = np.array([float(d.covariance.slog_pdet.get_value()
                           if hasattr(d.covariance.slog_pdet, "get_value")
                           else d.covariance.slog_pdet) for d in datasets])
    count = _Counter()
    hp = np.empty((C, n_t))
    for i, d in enumerate(datasets):
        name = get_hyper_name(d)
        h = np.asarray(hyperparams[name], dtype=np.float64)
        if hp_specific:
            h = h[..., count(name)]
        hp[:, i] = h if h.ndim else float(h)
    wid = ctx.weights_create_dense(W, slog)
    try:
        out = ctx.mvn_chol_logp_batch(wid, np.ascontiguousarray(res), hp)
    finally:
        ctx.weights_destroy(wid)
    return out if batched else out[0]
