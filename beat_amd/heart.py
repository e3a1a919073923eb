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
    ``beatamd_chol_inverse_batch`` (csrc/chol.hip): blocked factorisation of the exchange-flipped
    matrices on the FP64 matrix cores; numpy in, numpy out."""
    from .engine import get_context
    dev_index = None if device is None else getattr(device, "index", device)
    W, log_pdet = get_context(dev_index).chol_inverse_batch(np.ascontiguousarray(covs, dtype=np.float64))
    return W, log_pdet


# ------------------------------------------------------------------------------------------------
# Geometry-mode synthetics seams (beat/heart.py:3564-3762 seis_synthetics, :4158-4239 geo_synthetics)
#
# In the reference both are thin wrappers around ``engine.process(sources, targets)`` of pyrocko's
# GF-store engine (layered medium; not in the reference tree).  The arithmetic shipped here is the
# homogeneous half space (``HalfspaceEngine``: rectangular dislocations, Okada 1985, and Mogi
# sources on the GPU); results are NOT comparable with a layered GF store to 1e-6 -- parity with
# BEAT is unpinned for geometry mode, the kernels are pinned to Okada's published check values
# (tests/test_geometry.py).  What is reproduced exactly is the call protocol: argument order,
# ``outmode`` values, ordering and stacking of the per-(source, target) results.

km = 1000.0


class StaticTarget(object):
    """observation points of one geodetic dataset: east / north shifts [m] from the reference
    location (the fields of pyrocko.gf.StaticTarget this path reads)"""

    def __init__(self, east_shifts, north_shifts, lats=None, lons=None):
        self.east_shifts = np.ascontiguousarray(east_shifts, dtype=np.float64)
        self.north_shifts = np.ascontiguousarray(north_shifts, dtype=np.float64)
        self.lats = np.zeros_like(self.east_shifts) if lats is None else np.asarray(lats)
        self.lons = np.zeros_like(self.east_shifts) if lons is None else np.asarray(lons)


class HalfspaceSource(object):
    """attribute bag with pyrocko's RectangularSource field names (SI units: m, deg);
    kind "rectangular" | "mogi" (for Mogi: ``volume_change`` [m^3])"""

    _defaults = dict(east_shift=0.0, north_shift=0.0, depth=0.0, strike=0.0, dip=90.0, rake=0.0,
                     length=0.0, width=0.0, slip=0.0, opening_fraction=0.0, volume_change=0.0,
                     time=0.0)

    def __init__(self, kind="rectangular", **kwargs):
        if kind not in ("rectangular", "mogi"):
            raise ValueError("unknown source kind %s" % kind)
        self.kind = kind
        for k, v in self._defaults.items():
            setattr(self, k, float(kwargs.pop(k, v)))
        if kwargs:
            raise TypeError("unknown source attributes: %s" % ", ".join(sorted(kwargs)))

    def update(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, float(np.ravel(v)[0]))

    def kernel_parameters(self):
        """the 10 slots of beatamd_halfspace_displacements_batch ([km], [deg], [m])"""
        amp = self.volume_change if self.kind == "mogi" else self.slip
        return [self.east_shift / km, self.north_shift / km, self.depth / km, self.strike, self.dip,
                self.rake, self.length / km, self.width / km, amp, self.opening_fraction]


class HalfspaceEngine(object):
    """stands where the reference passes a pyrocko ``LocalEngine``: static displacements of
    half-space sources at surface points, evaluated on the GPU"""

    def __init__(self, nu=0.25, ctx=None):
        self.nu, self._ctx = float(nu), ctx

    @property
    def ctx(self):
        if self._ctx is None:
            from .engine import get_context
            self._ctx = get_context()
        return self._ctx

    def close_cashed_stores(self):   # the reference's Ops call this before pickling
        pass

    def __getstate__(self):
        return {"nu": self.nu}       # the device context is per process: looked up again on use

    def __setstate__(self, state):
        self.nu, self._ctx = state["nu"], None

    def static_displacements(self, sources, targets):
        """-> list, index i_t + i_s * n_targets (heart.py:4213-4216), of (n_points, 3) arrays
        [north, east, up] = [n, e, -d] (heart.py:4218-4224)"""
        kinds = [1 if s.kind == "mogi" else 0 for s in sources]
        prm = np.array([[s.kernel_parameters() for s in sources]])
        east = np.concatenate([t.east_shifts for t in targets]) / km
        north = np.concatenate([t.north_shifts for t in targets]) / km
        out = self.ctx.halfspace_displacements_batch(kinds, prm, east, north, self.nu)[0]
        bounds = np.cumsum([0] + [t.east_shifts.size for t in targets])
        return [out[i_s, bounds[i_t]:bounds[i_t + 1]].copy()
                for i_s in range(len(sources)) for i_t in range(len(targets))]


def geo_synthetics(engine, targets, sources, outmode="stacked_array", plot=False, nthreads=1):
    """heart.py:4158-4239 with the same arguments and ``outmode`` values:
    "arrays" one array per (source, target); "array" all of them stacked vertically;
    "stacked_arrays" per target, summed over the sources; "stacked_array" those stacked."""
    if not hasattr(engine, "static_displacements"):
        raise TypeError("engine %r provides no static displacements (HalfspaceEngine does; the "
                        "reference's layered GF-store engine is pyrocko's and out of scope)" % (engine,))
    disp_arrays = engine.static_displacements(list(sources), list(targets))
    ns, nt = len(sources), len(targets)

    def per_target():
        return [sum(disp_arrays[i_t + i_s * nt] for i_s in range(ns)) for i_t in range(nt)]

    if outmode == "arrays":
        return disp_arrays
    if outmode == "array":
        return np.vstack(disp_arrays)
    if outmode == "stacked_arrays":
        return per_target()
    if outmode == "stacked_array":
        return np.vstack(per_target())
    raise ValueError("Outmode %s not available" % outmode)


def seis_synthetics(engine, sources, targets, arrival_taper=None, wavename="any_P", filterer=None,
                    reference_taperer=None, plot=False, nthreads=1, outmode="array",
                    pre_stack_cut=False, taper_tolerance_factor=0.0, arrival_times=None,
                    chop_bounds=["b", "c"]):
    """heart.py:3564-3762 call protocol.  The waveform synthesis itself (``engine.process`` on a
    layered GF store, then filter / taper / chop through pyrocko ``Trace`` methods,
    heart.py:3658-3700) is third-party arithmetic outside the reference tree and is NOT
    re-implemented: ``engine`` must provide ``seismograms(sources, targets, ...) ->
    (traces (n_sources * n_targets, n_samples), tmins (n_targets,))`` already post-processed.  This
    function does what the reference does with them afterwards: the stack over sources
    (:3719-3724) and the ``outmode`` dispatch ("array", "data")."""
    if not hasattr(engine, "seismograms"):
        raise NotImplementedError(
            "geometry-mode seismic synthetics need a waveform engine (pyrocko GF stores in the "
            "reference); none is part of this package -- the finite-fault (FFI) path computes its "
            "synthetics from the linear GF library instead (beat_amd.ffi)")
    synths, tmins = engine.seismograms(sources, targets, arrival_taper=arrival_taper, wavename=wavename,
                                       filterer=filterer, arrival_times=arrival_times,
                                       chop_bounds=chop_bounds)
    synths = np.asarray(synths, dtype=np.float64)
    ns, nt = len(sources), len(targets)
    if synths.shape[0] != ns * nt:
        raise ValueError("Stacking error, traces different lengths!")
    outstack = synths if ns == 1 else sum(synths[k * nt:(k + 1) * nt] for k in range(ns))
    if outmode == "array":
        return outstack, np.asarray(tmins)
    if outmode == "data":
        return list(synths), np.asarray(tmins)
    raise TypeError("Outmode %s not supported!" % outmode)
