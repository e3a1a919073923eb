"""
Seeded synthetic distributed-slip problems: the shapes of
BASELINE.json's configurations with random-normal Green's functions.  Used by the
tests, ``__graft_entry__.smoke()`` and ``bench.py``; no reference data is needed.

Everything here is input construction -- no forward-model arithmetic.
"""
from collections import OrderedDict

import numpy as np

from .ffi import (GeodeticGFLibrary, GeodeticGFLibraryConfig, SeismicGFLibrary,
                  SeismicGFLibraryConfig)
from .models.problem import (FFIProblem, GeodeticData, ParameterLayout, SeismicWavemap,
                             hyper_name_laplacian)


def exponential_data_covariance(n, dt, tzero):
    """reference beat/covariance.py:24-51 (structure only)"""
    i = np.arange(n)
    return np.exp(-np.abs(i[:, None] - i[None, :]) * (dt / tzero))


def smoothing_operator_nearest_neighbor(n_patch_strike, n_patch_dip, ps_strike, ps_dip):
    """reference beat/models/laplacian.py:209-258 (5-point Laplacian, one-sided at the edges)"""
    n = n_patch_dip * n_patch_strike
    op = np.zeros((n, n))
    dd, ds = 1.0 / ps_dip ** 2, 1.0 / ps_strike ** 2
    for i in range(n):
        r, c = divmod(i, n_patch_strike)
        nb = [(r > 0, -n_patch_strike, dd), (r < n_patch_dip - 1, n_patch_strike, dd),
              (c > 0, -1, ds), (c < n_patch_strike - 1, 1, ds)]
        op[i, i] = -sum(w for ok, _, w in nb if ok)
        for ok, off, w in nb:
            if ok:
                op[i, i + off] = w
    return op


class SyntheticSpec(object):
    """Sizes and grids of a synthetic FFI problem.

    Defaults = BASELINE config 3: one 20x20 subfault of 1 km patches, 64 targets, 4096
    samples, D=3 durations from 0.5 s every 0.5 s, S=25 start times from 0 s every 0.5 s."""

    def __init__(self, n_patch_dip=(20,), n_patch_strike=(20,), patch_size=(1.0,), T=64, N=4096,
                 D=3, S=25, st_min=0.0, st_dt=0.5, du_min=0.5, du_dt=0.5,
                 slip_varnames=("uparr",), covariance="scalar", sigma=0.5, station_shifts=False,
                 geodetic_nobs=None, laplacian=False, interpolation="nearest_neighbor",
                 hp_specific=False, seed=20250711, vel_bounds=(2.5, 4.0), nuc_margin=0.0,
                 time_bounds=(0.0, 1.0)):
        self.n_patch_dip = tuple(n_patch_dip)
        self.n_patch_strike = tuple(n_patch_strike)
        self.patch_size = tuple(patch_size)
        self.T, self.N, self.D, self.S = T, N, D, S
        self.st_min, self.st_dt, self.du_min, self.du_dt = st_min, st_dt, du_min, du_dt
        self.slip_varnames = tuple(slip_varnames)
        self.covariance = covariance  # "scalar" | "toeplitz"
        self.sigma = sigma
        self.station_shifts = station_shifts
        self.geodetic_nobs = geodetic_nobs  # None or tuple of dataset sizes
        self.laplacian = laplacian
        self.interpolation = interpolation
        self.hp_specific = hp_specific
        self.seed = seed
        self.vel_bounds = vel_bounds
        self.nuc_margin = nuc_margin  # [km] keep the hypocentre this far from the fault edges
        self.time_bounds = time_bounds

    @property
    def nsub(self):
        return len(self.n_patch_dip)

    @property
    def P(self):
        return int(sum(d * s for d, s in zip(self.n_patch_dip, self.n_patch_strike)))

    @property
    def lib_bytes(self):
        return self.T * self.P * self.D * self.S * self.N * 8


def _layout_and_bounds(spec):
    P, nsub = spec.P, spec.nsub
    sizes, lower, upper = OrderedDict(), {}, {}

    def add(name, size, lo, up):
        sizes[name] = size
        lower[name], upper[name] = lo, up

    seismic = spec.T > 0
    for v in spec.slip_varnames:
        add(v, P, 0.0 if v == "uparr" else -1.0, 5.0 if v == "uparr" else 1.0)
    if seismic:
        add("durations", P, spec.du_min, spec.du_min + (spec.D - 1) * spec.du_dt)
        add("velocities", P, spec.vel_bounds[0], spec.vel_bounds[1])
        # per-subfault bounds (reference apps/beat.py:1577-1589 sets them from the fault
        # length / width); keep the rounded nucleation index inside the grid
        ext_s = np.array([s * h - 0.51 * h for s, h in zip(spec.n_patch_strike, spec.patch_size)])
        ext_d = np.array([d * h - 0.51 * h for d, h in zip(spec.n_patch_dip, spec.patch_size)])
        add("nucleation_strike", nsub, spec.nuc_margin, ext_s - spec.nuc_margin)
        add("nucleation_dip", nsub, spec.nuc_margin, ext_d - spec.nuc_margin)
        add("time", nsub, spec.time_bounds[0], spec.time_bounds[1])
        if spec.station_shifts:
            add("time_shifts_any_P_0", max(spec.T // 2, 1), -1.0, 1.0)
        add("h_any_P_0_Z", spec.T if spec.hp_specific else 1, -2.0, 2.0)
    if spec.geodetic_nobs:
        add("h_SAR", len(spec.geodetic_nobs) if spec.hp_specific else 1, -2.0, 2.0)
    if spec.laplacian:
        add(hyper_name_laplacian, 1, -2.0, 2.0)
    return ParameterLayout(sizes), lower, upper


def _one_pass_uniform_sweep(n_i, n_j, h, slow, hi, hj):
    """First outer iteration (four Gauss-Seidel sweeps in the order i+j+, i-j+, i-j-, i+j-) of the
    first-order Godunov upwind eikonal update on a grid of constant slowness -- the scheme the
    rupture-time kernel follows (reference fast_sweep_ext.c:65-206).  Input construction only:
    used to bound the start-time axis a synthetic library needs."""
    t = np.full((n_i, n_j), np.inf)
    t[hi, hj] = 0.0
    fh = slow * h
    orders = ((range(n_i), range(n_j)), (range(n_i - 1, -1, -1), range(n_j)),
              (range(n_i - 1, -1, -1), range(n_j - 1, -1, -1)), (range(n_i), range(n_j - 1, -1, -1)))
    for ri, rj in orders:
        for i in ri:
            for j in rj:
                a = min(t[i - 1, j] if i > 0 else np.inf, t[i + 1, j] if i < n_i - 1 else np.inf)
                b = min(t[i, j - 1] if j > 0 else np.inf, t[i, j + 1] if j < n_j - 1 else np.inf)
                if not (np.isfinite(a) or np.isfinite(b)):
                    continue
                if not np.isfinite(a) or not np.isfinite(b) or abs(a - b) >= fh:
                    cand = min(a, b) + fh
                else:
                    cand = 0.5 * (a + b + np.sqrt(2.0 * fh * fh - (a - b) ** 2))
                if cand < t[i, j]:
                    t[i, j] = cand
    return t


def max_sweep_time(spec):
    """Upper bound of the rupture onset time of ANY chain inside the prior box.

    Every Gauss-Seidel update of the upwind scheme is monotone in its neighbours and in the
    slowness, and the iterates only decrease from +inf; by induction over the updates the result
    of any number of outer iterations on slownesses <= 1/v_min is bounded cell by cell by the
    FIRST outer iteration on the homogeneous slowest medium with the same hypocentre.  That
    homogeneous time grows with the distance from the hypocentre, so the extreme hypocentres of
    the allowed range give the maximum.  (The Manhattan-distance bound used before is ~35 % larger
    and forced a narrow nucleation prior: config 3 gives 11.15 s here against 15.2 s, inside the
    12 s the S = 25 start-time axis of SURVEY 8(d) covers.)"""
    worst = 0.0
    slow = 1.0 / spec.vel_bounds[0]
    for nd, ns, h in zip(spec.n_patch_dip, spec.n_patch_strike, spec.patch_size):
        # nucleation positions [margin, n*h - 0.51h - margin] -> index range (positions2idxs)
        lo = int(np.rint((spec.nuc_margin - h / 2.0) / h))
        lo_i, lo_j = max(lo, 0), max(lo, 0)
        up_i = min(int(np.rint((nd * h - 0.51 * h - spec.nuc_margin - h / 2.0) / h)), nd - 1)
        up_j = min(int(np.rint((ns * h - 0.51 * h - spec.nuc_margin - h / 2.0) / h)), ns - 1)
        for hi in sorted({lo_i, max(up_i, lo_i)}):
            for hj in sorted({lo_j, max(up_j, lo_j)}):
                worst = max(worst, float(_one_pass_uniform_sweep(nd, ns, h, slow, hi, hj).max()))
    return worst


def draw_population(spec, layout, lower, upper, n_chains, seed_offset=1000):
    """chain c drawn from default_rng(seed_offset + c) uniformly inside the prior box
    (initialize_population, metropolis.py:125-152: prior draws)"""
    lo, up = layout.bounds(lower, upper)
    Q = np.empty((n_chains, layout.size))
    for c in range(n_chains):
        rng = np.random.default_rng(seed_offset + c)
        Q[c] = lo + (up - lo) * rng.random(layout.size)
    return Q


def build_problem(spec, device_library=False, ctx=None):
    """-> (FFIProblem, host_arrays dict for the oracle).

    device_library=True generates the seismic libraries directly in HBM with torch
    (for sizes that should not be materialised on the host, e.g. the 62.9 GB of config 3);
    host_arrays then holds no G."""
    rng = np.random.default_rng(spec.seed)
    layout, lower, upper = _layout_and_bounds(spec)
    P, T, N = spec.P, spec.T, spec.N
    host = dict(spec=spec, layout=layout, lower=lower, upper=upper)
    seismic = T > 0
    # make sure the library start-time axis covers sweep + time - shifts
    if seismic:
        need = max_sweep_time(spec) + spec.time_bounds[1] + (1.0 if spec.station_shifts else 0.0)
        have = spec.st_min + (spec.S - 1) * spec.st_dt
        if need > have + 1e-9:
            raise ValueError("library start-time axis (%.2f s) does not cover the rupture (%.2f s)"
                             % (have, need))

    wavemaps = []
    if seismic:
        gfs, Gs = {}, []
        for v in spec.slip_varnames:
            cfg = SeismicGFLibraryConfig(dimensions=(T, P, spec.D, spec.S, N),
                                         starttime_sampling=spec.st_dt,
                                         duration_sampling=spec.du_dt, starttime_min=spec.st_min,
                                         duration_min=spec.du_min, component=v)
            gf = SeismicGFLibrary(cfg)
            if device_library:
                import torch
                dev = torch.device("cuda", ctx.device if ctx is not None else 0)
                gen = torch.Generator(device=dev)
                gen.manual_seed(spec.seed + len(Gs))
                G = torch.empty((T, P, spec.D, spec.S, N), dtype=torch.float64, device=dev)
                flat = G.view(-1)
                step = 1 << 28
                for o in range(0, flat.numel(), step):
                    flat[o:o + step].normal_(generator=gen)
                gf.adopt_device_tensor(G)
                Gs.append(None)
            else:
                gf.setup(T, P, spec.D, spec.S, N, allocate=True)
                gf._gfmatrix[:] = rng.standard_normal((T, P, spec.D, spec.S, N))
                Gs.append(gf._gfmatrix)
            gfs[v] = gf
        data = spec.sigma * rng.standard_normal((T, N)) + rng.standard_normal((T, N)) * 3.0
        if spec.covariance == "scalar":
            sig = spec.sigma * (1.0 + 0.1 * rng.random(T))
            weights = 1.0 / sig                       # chol_inverse of sigma^2 I = I / sigma
            slog = N * np.log(sig ** 2)               # log det(sigma^2 I)
        else:
            # covariance.py:413-427: C_i = scaling_i * structure(dt, tzero)
            base = exponential_data_covariance(N, 0.5, 2.0)
            Lb = np.linalg.cholesky(base)
            Wb = np.linalg.cholesky(np.linalg.inv(base)).T
            ldb = 2.0 * np.log(np.diag(Lb)).sum()
            scal = (spec.sigma * (1.0 + 0.1 * rng.random(T))) ** 2
            weights = np.stack([Wb / np.sqrt(s) for s in scal])
            slog = np.array([ldb + N * np.log(s) for s in scal])
        hypers = [("h_any_P_0_Z", t if spec.hp_specific else 0) for t in range(T)]
        ts = None
        if spec.station_shifts:
            nst = layout.varsizes["time_shifts_any_P_0"]
            ts = ("time_shifts_any_P_0", np.arange(T) % nst)
        wavemaps.append(SeismicWavemap(gfs, data, weights, slog, hypers, ts, spec.interpolation))
        host.update(Gs=Gs, data=data, weights=weights, slog=slog, hypers=hypers, time_shifts=ts)

    geodetic = None
    if spec.geodetic_nobs:
        nobs = int(sum(spec.geodetic_nobs))
        ggfs, gGs = {}, []
        for v in spec.slip_varnames:
            gg = GeodeticGFLibrary(GeodeticGFLibraryConfig(dimensions=(P, nobs), component=v))
            gg.setup(P, nobs, allocate=True)
            gg._gfmatrix[:] = 1e-1 * rng.standard_normal((P, nobs))
            ggfs[v] = gg
            gGs.append(gg._gfmatrix)
        gdata = rng.standard_normal(nobs)
        godw = 0.5 + rng.random(nobs)
        gW, gslog = [], []
        for n in spec.geodetic_nobs:
            b = rng.standard_normal((n, n))
            Cg = b @ b.T / n + np.eye(n)
            gW.append(np.linalg.cholesky(np.linalg.inv(Cg)).T)
            gslog.append(2.0 * np.log(np.diag(np.linalg.cholesky(Cg))).sum())
        ghyp = [("h_SAR", k if spec.hp_specific else 0) for k in range(len(spec.geodetic_nobs))]
        geodetic = GeodeticData(ggfs, gdata, godw, spec.geodetic_nobs, gW, gslog, ghyp)
        host.update(gGs=gGs, gdata=gdata, godw=godw, gW=gW, gslog=gslog, ghyp=ghyp)

    lap = None
    if spec.laplacian:
        if spec.nsub != 1:
            raise ValueError("nearest-neighbour smoothing operator is defined for one subfault")
        L = smoothing_operator_nearest_neighbor(spec.n_patch_strike[0], spec.n_patch_dip[0],
                                                spec.patch_size[0], spec.patch_size[0])
        LtL = L.T * L  # elementwise, as in laplacian.py:58
        logdet = 2.0 * np.log(np.diag(np.linalg.cholesky(LtL))).sum()
        lap = (L, logdet)
        host.update(L=L, lap_logdet=logdet)

    prob = FFIProblem(layout, spec.n_patch_dip, spec.n_patch_strike, spec.patch_size,
                      spec.slip_varnames, wavemaps, geodetic, lap, lower, upper)
    return prob, host


def build_geometry_problem(sizes=(214, 205), seed=11):
    """BASELINE configs[1] shape on synthetic inputs: one rectangular source in a homogeneous half
    space (nine sampled source parameters with the bounds of the reference's
    data/examples/Fernandina/config_geometry.yaml:26-99 in spirit, one hyper-parameter), two SAR
    scenes of `sizes` points with full covariances (exponential in the point distance + a nugget),
    random observation geometry and line-of-sight angles in the range of an ascending / descending
    pair.  -> (GeodeticGeometryProblem, layout, lower, upper).  Input construction only; the
    geometry-mode forward model itself has no BEAT-anchored parity check (pyrocko is not in the
    reference tree -- DESIGN.md section 4)."""
    from .heart import whitening
    from .models import GeodeticGeometryProblem, ParameterLayout, los_vectors
    rng = np.random.default_rng(seed)
    nobs = int(sum(sizes))
    east, north = rng.uniform(-15, 15, nobs), rng.uniform(-15, 15, nobs)
    inc = np.concatenate([rng.uniform(33, 43, n) if i % 2 == 0 else rng.uniform(20, 28, n)
                          for i, n in enumerate(sizes)])
    head = np.concatenate([np.full(n, -12.0 if i % 2 == 0 else -168.0) + rng.normal(0, 0.3, n)
                           for i, n in enumerate(sizes)])
    names = OrderedDict([("depth", 1), ("dip", 1), ("east_shift", 1), ("length", 1), ("north_shift", 1),
                         ("slip", 1), ("strike", 1), ("width", 1), ("h_SAR", 1)])
    lay = ParameterLayout(names)
    lower = dict(depth=0.5, dip=5.0, east_shift=-5.0, length=0.5, north_shift=-5.0, slip=0.01, strike=0.0,
                 width=0.5, h_SAR=-2.0)
    upper = dict(depth=9.0, dip=85.0, east_shift=5.0, length=10.0, north_shift=5.0, slip=1.0, strike=360.0,
                 width=8.0, h_SAR=2.0)
    data = 0.01 * rng.standard_normal(nobs)
    odw = 0.5 + rng.random(nobs)
    Ws, sls, o = [], [], 0
    for n in sizes:
        x, y = east[o:o + n], north[o:o + n]
        dist = np.hypot(x[:, None] - x[None, :], y[:, None] - y[None, :])
        C = 1e-4 * (np.exp(-dist / 5.0) + 0.1 * np.eye(n))
        W, sl = whitening(C)
        Ws.append(W)
        sls.append(sl)
        o += n
    prob = GeodeticGeometryProblem(lay, ["rectangular"], east, north, los_vectors(inc, head), data, odw, sizes,
                                   Ws, sls, [("h_SAR", 0)] * len(sizes), fixed=dict(rake=30.0, opening_fraction=0.25),
                                   lower=lower, upper=upper)
    return prob, lay, lower, upper
