"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the BEAT SMC/PT hot path.

ctypes front-end of ``oracle/libbeat_oracle.so`` (beat_oracle.c, the plain-C
restatement) plus numpy restatements of the per-stage linear algebra
(covariance factorisation, smoothing operators, weighted covariance).  Every
function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``beat_amd/``) never does.

Pinned by: tests/test_oracle_golden.py (reference numpy path golden vectors),
tests/test_oracle_kat.py (the reference test-suite's known-answer tests) and
tests/test_oracle_ref.py (oracle/_ref = reference C extension compiled in place).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NN, ML = 0, 1
_INTERP = {"nearest_neighbor": NN, "multilinear": ML}

_dp = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libbeat_oracle.so")
    src = os.path.join(_HERE, "beat_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbeat_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.bo_mvn_chol_logp.restype = C.c_double
        _LIB.bo_mvn_diag_logp.restype = C.c_double
        _LIB.bo_laplacian_logp.restype = C.c_double
        _LIB.bo_calc_beta.restype = C.c_double
        _LIB.bo_pt_tune.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ------------------------------------------------------------------ C restatements
def positions2idxs(positions, cell_size, min_pos=0.0):
    """utility.py:1542-1558"""
    pos = _f64(np.atleast_1d(positions))
    out = np.empty(pos.shape, dtype=np.int16)
    lib().bo_positions2idxs(_p(pos), C.c_long(pos.size), C.c_double(cell_size),
                            C.c_double(min_pos), out.ctypes.data_as(C.c_void_p))
    return out


def fast_sweep(slowness, patch_size, h_strk, h_dip, num_strk, num_dip):
    """fast_sweep_ext.c:120-245; same argument order as the reference extension."""
    s = _f64(slowness).ravel()
    assert s.size == num_strk * num_dip
    if not (0 <= h_strk < num_strk and 0 <= h_dip < num_dip):
        raise IndexError("hypocentre index outside the grid")
    out = np.empty(s.size)
    lib().bo_fast_sweep(_p(s), C.c_double(patch_size), C.c_long(int(h_strk)),
                        C.c_long(int(h_dip)), C.c_long(int(num_strk)), C.c_long(int(num_dip)),
                        _p(out))
    return out


def time2idx(x, xmin, dx, interpolation="nearest_neighbor"):
    """ffi/base.py:486-568 (starttimes2idxs / durations2idxs)"""
    x = _f64(x)
    idx = np.empty(x.shape, dtype=np.int16)
    if interpolation == "nearest_neighbor":
        lib().bo_time2idx_nn(_p(x), C.c_long(x.size), C.c_double(xmin), C.c_double(dx),
                             idx.ctypes.data_as(C.c_void_p))
        return idx, None
    fac = np.empty(x.shape)
    lib().bo_time2idx_ml(_p(x), C.c_long(x.size), C.c_double(xmin), C.c_double(dx),
                         idx.ctypes.data_as(C.c_void_p), _p(fac))
    return idx, fac


def stack_all(G, durations, starttimes, slips, dur_min, dur_dt, st_min, st_dt,
              interpolation="nearest_neighbor"):
    """ffi/base.py:607-709.  G (T,P,D,S,N); starttimes (T,P); -> (T,N)"""
    G = _f64(G)
    T, P, D, S, N = G.shape
    d = _f64(durations).ravel()
    st = _f64(np.broadcast_to(starttimes, (T, P)))
    sl = _f64(slips).ravel()
    out = np.empty((T, N))
    rc = lib().bo_stack_all(_p(G), C.c_long(T), C.c_long(P), C.c_long(D), C.c_long(S),
                            C.c_long(N), _p(d), _p(st), _p(sl), C.c_int(_INTERP[interpolation]),
                            C.c_double(dur_min), C.c_double(dur_dt), C.c_double(st_min),
                            C.c_double(st_dt), _p(out))
    if rc != 0:
        raise IndexError("index out of bounds of the GF library")
    return out


def geo_stack(G, slips):
    """ffi/base.py:292-305  G (P,Nobs)"""
    G = _f64(G)
    s = _f64(slips).ravel()
    out = np.empty(G.shape[1])
    lib().bo_geo_stack(_p(G), C.c_long(G.shape[0]), C.c_long(G.shape[1]), _p(s), _p(out),
                       C.c_int(0))
    return out


def mvn_chol_logp(W, r, slog_pdet, hp):
    """distributions.py:119-138, one dataset; W dense (M,M) or scalar"""
    r = _f64(r).ravel()
    if np.ndim(W) == 0:
        return lib().bo_mvn_diag_logp(C.c_double(float(W)), _p(r), C.c_long(r.size),
                                      C.c_double(slog_pdet), C.c_double(hp), None)
    W = _f64(W)
    return lib().bo_mvn_chol_logp(_p(W), _p(r), C.c_long(r.size), C.c_double(slog_pdet),
                                  C.c_double(hp), None)


def multivariate_normal_chol(weights, slog_pdets, hps, residuals):
    """distributions.py:72-140 over a list of datasets (hp already resolved per dataset)"""
    return np.array([mvn_chol_logp(W, r, sl, hp)
                     for W, r, sl, hp in zip(weights, residuals, slog_pdets, hps)])


def laplacian_logp(L, s, logdet, hp):
    """laplacian.py:88-134, one slip variable"""
    L = _f64(L)
    s = _f64(s).ravel()
    return lib().bo_laplacian_logp(_p(L), C.c_long(s.size), _p(s), C.c_double(logdet),
                                   C.c_double(hp))


def ffi_seismic_forward(Gs, lib_cfg, fault, params, data, weights, slog_pdet, hp,
                        time_shifts=None, interpolation="nearest_neighbor",
                        return_synthetics=True):
    """seismic.py:1253-1341 for ONE chain.

    Gs        list of (T,P,D,S,N) libraries, one per slip variable
    lib_cfg   dict(dur_min, dur_dt, st_min, st_dt)
    fault     dict(ndip=[..], nstrike=[..], patch_size=[..]) per subfault
    params    dict(slips (nvar,P), durations (P), velocities (P), nuc_strike (nsub),
                   nuc_dip (nsub), time (nsub))
    weights   (T,) scalars or (T,N,N) dense chol_inverse
    returns   starttimes0 (P), synthetics (T,N) or None, logpts (T)
    """
    Gs = [_f64(G) for G in Gs]
    T, P, D, S, N = Gs[0].shape
    nvar = len(Gs)
    Gp = (_dp * nvar)(*[_p(G) for G in Gs])
    nsub = len(fault["ndip"])
    ndip = (C.c_long * nsub)(*[int(v) for v in fault["ndip"]])
    nstr = (C.c_long * nsub)(*[int(v) for v in fault["nstrike"]])
    psz = _f64(fault["patch_size"])
    slips = _f64(params["slips"]).reshape(nvar, P)
    dur = _f64(params["durations"])
    vel = _f64(params["velocities"])
    ns = _f64(np.atleast_1d(params["nuc_strike"]))
    nd = _f64(np.atleast_1d(params["nuc_dip"]))
    t0 = _f64(np.atleast_1d(params["time"]))
    data = _f64(data)
    weights = _f64(weights)
    wkind = 0 if weights.ndim == 1 else 1
    slog = _f64(slog_pdet)
    hp = _f64(np.broadcast_to(hp, (T,)))
    ts = None if time_shifts is None else _f64(time_shifts)
    st0 = np.empty(P)
    syn = np.empty((T, N)) if return_synthetics else None
    logpts = np.empty(T)
    rc = lib().bo_ffi_seismic_forward(
        Gp, C.c_long(nvar), C.c_long(T), C.c_long(P), C.c_long(D), C.c_long(S), C.c_long(N),
        C.c_double(lib_cfg["dur_min"]), C.c_double(lib_cfg["dur_dt"]),
        C.c_double(lib_cfg["st_min"]), C.c_double(lib_cfg["st_dt"]),
        C.c_int(_INTERP[interpolation]),
        C.c_long(nsub), ndip, nstr, _p(psz),
        _p(slips), _p(dur), _p(vel), _p(ns), _p(nd), _p(t0),
        None if ts is None else _p(ts),
        _p(data), C.c_int(wkind), _p(weights), _p(slog), _p(hp),
        _p(st0), None if syn is None else _p(syn), _p(logpts))
    if rc != 0:
        raise IndexError("index out of bounds of the GF library")
    return st0, syn, logpts


def metrop_accept(beta, like_prop, like_prev, log_u):
    """metropolis.py:355-358 + pymc metrop_select semantics"""
    return bool(lib().bo_metrop_accept(C.c_double(beta), C.c_double(like_prop),
                                       C.c_double(like_prev), C.c_double(log_u)))


def calc_beta(likelihoods, beta, coef_variation=1.0):
    """smc.py:133-165 -> (beta_new, old_beta, weights)"""
    lk = _f64(likelihoods).ravel()
    w = np.empty(lk.size)
    b = lib().bo_calc_beta(_p(lk), C.c_long(lk.size), C.c_double(beta),
                           C.c_double(coef_variation), _p(w))
    return b, beta, w


def resample(weights, aux):
    """smc.py:290-324 with aux = the single np.random.rand(1) draw"""
    w = _f64(weights).ravel()
    out = np.empty(w.size, dtype=np.int64)
    lib().bo_resample(_p(w), C.c_long(w.size), C.c_double(float(aux)),
                      out.ctypes.data_as(C.c_void_p))
    return out


def pt_swap_accept(beta1, beta2, llk1, llk2, log_u):
    """pt.py:429-457"""
    return bool(lib().bo_pt_swap_accept(C.c_double(beta1), C.c_double(beta2),
                                        C.c_double(llk1), C.c_double(llk2), C.c_double(log_u)))


def pt_tune(scale, acc_rate):
    """pt.py:37-73"""
    return lib().bo_pt_tune(C.c_double(scale), C.c_double(acc_rate))


def smc_tune(acc_rate):
    """smc.py:558-575 (Muto & Beck 2008)"""
    a, b = 1.0 / 9, 8.0 / 9
    return np.power((a + (b * acc_rate)), 2)


# --------------------------------------------------------- numpy restatements (setup)
def exponential_data_covariance(n, dt, tzero):
    """covariance.py:24-51"""
    i = np.arange(n)
    return np.exp(-np.abs(i[:, None] - i[None, :]) * (dt / tzero))


def cov_chol(Cx):
    """heart.py:201-209 Covariance.chol (lower)"""
    from scipy import linalg
    return linalg.cholesky(Cx, lower=True)


def cov_chol_inverse(Cx):
    """heart.py:211-237 Covariance.chol_inverse: cholesky(inv(C)).T, QR fallback"""
    try:
        return np.linalg.cholesky(np.linalg.inv(Cx)).T
    except np.linalg.LinAlgError:
        inverse_chol = np.linalg.inv(cov_chol(Cx).T)
        _, chol_ur = np.linalg.qr(inverse_chol.T)
        return chol_ur


def cov_log_pdet(Cx):
    """heart.py:239-245 Covariance.log_pdet"""
    return np.log(np.diag(cov_chol(Cx))).sum() * 2.0


def log_determinant(A, inverse=False):
    """heart.py:65-89"""
    from scipy import linalg
    ch = linalg.cholesky(A, lower=True)
    if inverse:
        ch = np.linalg.inv(ch)
    return np.log(np.diag(ch)).sum() * 2.0


def smoothing_operator_nearest_neighbor(n_patch_strike, n_patch_dip, patch_size_strike,
                                        patch_size_dip):
    """laplacian.py:180-258 (operator) -- 5-point Laplacian with one-sided edges"""
    n = n_patch_dip * n_patch_strike
    op = np.zeros((n, n))
    dl_dip = 1.0 / patch_size_dip ** 2
    dl_str = 1.0 / patch_size_strike ** 2
    for i in range(n):
        row, col = divmod(i, n_patch_strike)
        up, down = row > 0, row < n_patch_dip - 1
        left, right = col > 0, col < n_patch_strike - 1
        op[i, i] = -1 * (up * dl_dip + down * dl_dip + left * dl_str + right * dl_str)
        if up:
            op[i, i - n_patch_strike] = dl_dip
        if down:
            op[i, i + n_patch_strike] = dl_dip
        if left:
            op[i, i - 1] = dl_str
        if right:
            op[i, i + 1] = dl_str
    return op


def weighted_covariance(population, weights):
    """smc.py:167-186 calc_covariance (before ensure_cov_psd):
    np.cov(X, aweights=w, bias=False, rowvar=0)"""
    X = np.asarray(population, dtype=np.float64)
    w = np.asarray(weights, dtype=np.float64).ravel()
    v1 = w.sum()
    mean = (X * w[:, None]).sum(0) / v1
    Xc = X - mean
    fact = v1 - (w * w).sum() / v1
    return (Xc * w[:, None]).T @ Xc / fact


def los_vectors(incidence_deg, heading_deg):
    """heart.py:1381-1410 DiffIFG.update_los_vector: [Sn, Se, Su]"""
    inc = np.deg2rad(np.asarray(incidence_deg, dtype=np.float64))
    head = np.deg2rad(np.asarray(heading_deg, dtype=np.float64) - 270)
    Su = np.cos(inc)
    Sn = -np.sin(inc) * np.cos(head)
    Se = -np.sin(inc) * np.sin(head)
    return np.array([Sn, Se, Su], dtype=np.float64).T


def ffi_geodetic_logp(Gs, slips, data, odws, splits, weights, slog_pdets, hps):
    """geodetic.py:1065-1081 for ONE chain:
    mu = sum_var G_var.T @ slip_var ; residual = (data - mu) * odw, split per
    dataset (utility.py:329-348 srmap) ; multivariate_normal_chol"""
    mu = np.zeros(Gs[0].shape[1])
    for G, s in zip(Gs, slips):
        mu += geo_stack(G, s)
    res = (np.asarray(data) - mu) * np.asarray(odws)
    out = []
    o = 0
    for n, W, sl, hp in zip(splits, weights, slog_pdets, hps):
        out.append(mvn_chol_logp(W, res[o:o + n], sl, hp))
        o += n
    return np.array(out), mu


def running_window_rms(data, window_size, mode="valid"):
    """utility.py:1141-1161"""
    data2 = np.power(data, 2)
    window = np.ones(window_size) / float(window_size)
    return np.sqrt(np.convolve(data2, window, mode))


def autocovariance(data):
    """covariance.py:716-736"""
    d = _f64(data).ravel()
    out = np.empty(d.size)
    lib().bo_autocovariance(_p(d), C.c_long(d.size), C.c_double(d.mean()), _p(out))
    return out


def non_toeplitz_covariance(data, window_size):
    """covariance.py:739-771"""
    d = _f64(data).ravel()
    stds = running_window_rms(d, window_size=window_size, mode="same")
    coeffs = autocovariance(d / stds)
    out = np.empty((d.size, d.size))
    lib().bo_scaled_toeplitz(_p(coeffs), _p(_f64(stds)), C.c_long(d.size), _p(out))
    return out
