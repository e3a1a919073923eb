"""
Geometry-mode geodetic problem: rectangular dislocations / Mogi sources in a homogeneous half
space, the counterpart of ``GeodeticGeometryComposite`` (beat/models/geodetic.py:565-760) with
``RectangularSource`` parameters (beat/sources.py; priors e.g.
data/examples/Fernandina/config_geometry.yaml:26-99) for BASELINE configs 1 and 2.

In the reference the displacements come from pyrocko's GF-store engine
(``heart.geo_synthetics``, layered medium, not in its tree): results of this analytic engine
are NOT comparable to BEAT's to 1e-6 -- parity is unpinned there and pinned instead to Okada's
(1985) published check values.  LOS projection, residual weighting,
``multivariate_normal_chol``, hyper-parameters, the prior box and the Metropolis step are the
same code path as the FFI problem.
"""
import numpy as np

from .. import _lib
from ..engine import get_context
from .problem import LogpForwFunc

# parameter slots of one source, in the order the kernel reads them
SOURCE_PARAMS = ("east_shift", "north_shift", "depth", "strike", "dip", "rake", "length", "width",
                 "slip", "opening_fraction")
KIND = {"rectangular": 0, "mogi": 1}


def los_vectors(incidence_deg, heading_deg):
    """heart.py:1381-1410 DiffIFG.update_los_vector -> (n, 3) = [Sn, Se, Su]"""
    inc = np.deg2rad(np.asarray(incidence_deg, dtype=np.float64))
    head = np.deg2rad(np.asarray(heading_deg, dtype=np.float64) - 270)
    Su = np.cos(inc)
    Sn = -np.sin(inc) * np.cos(head)
    Se = -np.sin(inc) * np.sin(head)
    return np.array([Sn, Se, Su], dtype=np.float64).T


class GeodeticGeometryProblem(object):
    """
    layout    ParameterLayout over the sampled variables; source variables have one entry per
              source (``depth`` of size n_sources, ...), like BEAT's geometry problems
    sources   list of "rectangular" / "mogi"
    fixed     dict name -> value(s) for parameters that are not sampled (lower == upper in BEAT)
    east, north [km], los (Nobs, 3); data, odws (Nobs,); sizes per dataset;
    weights   list of (n_k, n_k) chol_inverse matrices or scalars; slog_pdets; hypers as in
              FFIProblem; for a Mogi source the volume change [m^3] is the ``slip`` slot
    """

    def __init__(self, layout, sources, east, north, los, data, odws, sizes, weights, slog_pdets,
                 hypers, fixed=None, nu=0.25, lower=None, upper=None):
        self.layout = layout
        self.sources = list(sources)
        self.east = np.ascontiguousarray(east, dtype=np.float64)
        self.north = np.ascontiguousarray(north, dtype=np.float64)
        self.los = np.ascontiguousarray(los, dtype=np.float64)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.odws = np.ascontiguousarray(odws, dtype=np.float64)
        self.sizes = [int(s) for s in sizes]
        self.weights, self.slog_pdets, self.hypers = weights, list(slog_pdets), list(hypers)
        self.fixed = dict(fixed or {})
        self.nu = float(nu)
        self.lower, self.upper = lower, upper
        # what LogpForwFunc expects of a problem
        self.wavemaps, self.geodetic, self.laplacian = [], self, None

    @property
    def out_names(self):
        return ["geo_like_%d" % i for i in range(len(self.sizes))] + ["like"]

    def source_tables(self):
        ns = len(self.sources)
        off = -np.ones((ns, len(SOURCE_PARAMS)), dtype=np.int64)
        fix = np.zeros((ns, len(SOURCE_PARAMS)))
        for s in range(ns):
            for k, name in enumerate(SOURCE_PARAMS):
                if name in self.layout.offsets:
                    off[s, k] = self.layout.offset(name, s if self.layout.varsizes[name] > 1 else 0)
                elif name in self.fixed:
                    fix[s, k] = np.atleast_1d(self.fixed[name])[min(s, np.size(self.fixed[name]) - 1)]
                elif name in ("opening_fraction", "rake"):
                    fix[s, k] = 0.0
                elif self.sources[s] == "mogi" and name in ("strike", "dip", "length", "width"):
                    fix[s, k] = 0.0
                else:
                    raise KeyError("source parameter %s is neither sampled nor fixed" % name)
        return off, fix

    def compile(self, ctx=None):
        ctx = ctx or get_context()
        L = _lib.FfiLayout()
        L.nparams = self.layout.size
        L.nvar = 0
        for i in range(4):
            L.slip_off[i] = -1
        for f in ("durations_off", "velocities_off", "nuc_strike_off", "nuc_dip_off", "time_off",
                  "h_laplacian_off"):
            setattr(L, f, -1)
        mid = ctx.ffi_model_create(L, [], [], [])
        self._wsets = []
        for n, W, sl in zip(self.sizes, self.weights, self.slog_pdets):
            if np.ndim(W) == 0:
                self._wsets.append(ctx.weights_create_scalar([float(W)], [sl], n))
            else:
                self._wsets.append(ctx.weights_create_dense(np.asarray(W), [sl]))
        off, fix = self.source_tables()
        hp_off = [self.layout.offset(n, i) for n, i in self.hypers]
        ctx.ffi_model_add_geodetic_geometry(mid, [KIND[s] for s in self.sources], off, fix, self.east,
                                            self.north, self.los, self.nu, self.data, self.odws,
                                            self.sizes, self._wsets, hp_off)
        return LogpForwFunc(ctx, mid, self)
