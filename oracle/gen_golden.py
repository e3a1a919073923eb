"""
TEST INFRASTRUCTURE ONLY.

Generates tests/golden/*.npz by running the REFERENCE's own code (numpy path +
its C extension compiled in place as oracle/_ref) on seeded inputs.  Needs
/root/reference, so it runs in the build container only; the fixtures it writes
are committed (they are data: inputs and expected outputs, no reference source).

    make -C oracle ref && python oracle/gen_golden.py

Reference entry points exercised (file:line under /root/reference):
  beat/fast_sweeping/fast_sweep_ext.c:120-245  fast_sweep            (C ext, in place)
  beat/fast_sweeping/fast_sweep.py:67-230      get_rupture_times_numpy
  beat/utility.py:1542-1558                    positions2idxs
  beat/ffi/base.py:486-568, 607-709            SeismicGFLibrary idx maps + stack_all
  beat/ffi/base.py:292-305                     GeodeticGFLibrary.stack_all
  beat/heart.py:65-89, 104-263                 log_determinant, Covariance
  beat/covariance.py:24-51, 716-771            exponential_data_covariance, autocovariance,
                                               non_toeplitz_covariance; utility.py:1141-1161
  beat/models/laplacian.py:209-258             get_smoothing_operator_nearest_neighbor
  beat/sampler/smc.py:133-186, 290-324, 558-575  calc_beta, np.cov weights, resample, tune
  beat/sampler/pt.py:37-73                     tune
  beat/utility.py:1034-1138                    ensure_cov_psd
  data/examples/Laquila/geodetic_data.pkl      (arrays only)
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()

import fast_sweep_ext  # noqa: E402  (oracle/_ref, the reference's C built in place)
from beat import covariance as rcov  # noqa: E402
from beat import heart, utility  # noqa: E402
from beat.config import GeodeticGFLibraryConfig, SeismicGFLibraryConfig  # noqa: E402
from beat.fast_sweeping import fast_sweep as rfs  # noqa: E402
from beat.ffi import base as ffibase  # noqa: E402
from beat.models import laplacian as rlap  # noqa: E402
from beat.sampler import pt as rpt  # noqa: E402
from beat.sampler import smc as rsmc  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f kB)" % (path, os.path.getsize(path) / 1e3))


# ---------------------------------------------------------------- fast sweep
def gen_sweep():
    rng = np.random.default_rng(20250711)
    cases = {}
    # reference test/test_fastsweep.py:27-31: 6 dip x 4 strike, nuc_x(strike)=2,
    # nuc_y(dip)=3, velocities 1.0 | 3.5, patch 10 km
    velo = np.concatenate((np.ones((6, 2)), np.ones((6, 2)) * 3.5), axis=1)
    specs = [("kat", 1.0 / velo, 10.0, 3, 2)]
    shapes = [(2, 2), (1, 7), (7, 1), (5, 9), (20, 20), (13, 17), (20, 10), (3, 30), (32, 24)]
    for k, (nd, ns) in enumerate(shapes):
        vel = rng.uniform(2.5, 4.0, (nd, ns))
        specs.append(("rand%d" % k, 1.0 / vel, float(rng.choice([0.5, 1.0, 2.0, 2.5])),
                      int(rng.integers(0, nd)), int(rng.integers(0, ns))))
    # homogeneous medium, corner / centre hypocentres
    specs.append(("homog_corner", np.full((20, 20), 1 / 3.5), 1.0, 0, 0))
    specs.append(("homog_corner2", np.full((20, 20), 1 / 3.5), 1.0, 19, 19))
    specs.append(("homog_centre", np.full((11, 11), 0.4), 2.0, 5, 5))
    # strong contrast (shadow zones exercise the one-sided branch)
    vel = rng.uniform(0.5, 6.0, (16, 16))
    specs.append(("contrast", 1.0 / vel, 1.0, 7, 3))
    names = []
    for name, slow, psz, hd, hs in specs:
        nd, ns = slow.shape
        c = fast_sweep_ext.fast_sweep(np.ascontiguousarray(slow).ravel(), psz, hd, hs, nd, ns)
        # numpy twin: (Slowness 2d, patch_size, n_patch_strike, n_patch_dip, nuc_x, nuc_y)
        n = rfs.get_rupture_times_numpy(slow, psz, ns, nd, hs, hd).ravel()
        cases[name + "_slow"] = slow
        cases[name + "_meta"] = np.array([psz, hd, hs, nd, ns], dtype=np.float64)
        cases[name + "_c"] = c
        cases[name + "_numpy"] = n
        names.append(name)
    cases["names"] = np.array(names)
    save("sweep", **cases)


def gen_sweep_ties():
    """slowness fields whose rupture times land EXACTLY on k + 0.5 ties of the
    start-time grid (t / dt = k + 0.5 with dt = 0.5, 0.25), where round-half-even decides the
    int16 index.  Slowness x patch size are dyadic numbers (0.25, 0.125, 0.75 ...), so along rows
    and columns through the hypocentre the one-sided updates are exact sums and every
    implementation that keeps the reference's operation order reproduces the tie bit for bit; a
    contracted or re-associated update would move it off the tie and flip the index."""
    cases, names = {}, []
    specs = []
    specs.append(("tie_homog_quarter", np.full((9, 12), 0.25), 1.0, 4, 5))       # t = 0.25 k
    specs.append(("tie_homog_075", np.full((8, 8), 0.375), 2.0, 0, 0))           # t = 0.75 k
    het = np.full((10, 10), 0.25)
    het[:, 5:] = 0.125                                                            # two media
    het[3, :] = 0.75
    specs.append(("tie_two_media", het, 1.0, 3, 2))
    stripes = np.tile(np.array([0.125, 0.375, 0.25, 0.625]), (12, 5))             # 12 x 20
    specs.append(("tie_stripes", stripes, 2.0, 6, 0))
    specs.append(("tie_eighth", np.full((20, 20), 0.125), 1.0, 19, 0))            # t = 0.125 k
    n_ties = 0
    for name, slow, psz, hd, hs in specs:
        nd, ns = slow.shape
        c = fast_sweep_ext.fast_sweep(np.ascontiguousarray(slow).ravel(), psz, hd, hs, nd, ns)
        cases[name + "_slow"] = slow
        cases[name + "_meta"] = np.array([psz, hd, hs, nd, ns], dtype=np.float64)
        cases[name + "_c"] = c
        for dt in (0.5, 0.25):
            x = c / dt
            tie = (x - np.floor(x)) == 0.5
            n_ties += int(tie.sum())
            # the reference's index maps on the reference's times (ffi/base.py:506-517)
            cases[name + "_idx_nn_%g" % dt] = np.round((c - 0.0) / dt).astype("int16")
            cases[name + "_idx_ml_%g" % dt] = np.ceil((c - 0.0) / dt).astype("int16")
            cases[name + "_ties_%g" % dt] = tie
        names.append(name)
    assert n_ties >= 50, n_ties
    cases["names"] = np.array(names)
    save("sweep_ties", **cases)


# ---------------------------------------------------------------- positions2idxs
def gen_positions():
    rng = np.random.default_rng(7)
    pos = np.concatenate([rng.uniform(0, 20, 64),
                          np.array([0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.5, 19.5, 19.999, 20.0])])
    out = {}
    for cell in (1.0, 2.0, 2.5):
        out["idx_%g" % cell] = utility.positions2idxs(pos, cell)
    out["pos"] = pos
    save("positions2idxs", **out)


# ---------------------------------------------------------------- gf stacking
def make_seis_lib(T, P, D, S, N, st_min, st_dt, du_min, du_dt, rng):
    cfg = SeismicGFLibraryConfig(dimensions=(T, P, D, S, N), starttime_sampling=st_dt,
                                 duration_sampling=du_dt, starttime_min=st_min,
                                 duration_min=du_min)
    gfs = ffibase.SeismicGFLibrary(config=cfg)
    gfs.setup(T, P, D, S, N, allocate=True)
    gfs._gfmatrix[:] = rng.standard_normal((T, P, D, S, N))
    gfs._stack_switch = {"numpy": gfs._gfmatrix}
    return gfs


def gen_stack():
    rng = np.random.default_rng(20250711)
    T, P, D, S, N = 5, 12, 4, 9, 24
    st_min, st_dt, du_min, du_dt = 0.0, 0.5, 0.5, 0.25
    gfs = make_seis_lib(T, P, D, S, N, st_min, st_dt, du_min, du_dt, rng)
    tidx = np.atleast_2d(np.arange(T)).T
    out = dict(G=gfs._gfmatrix, cfg=np.array([st_min, st_dt, du_min, du_dt]))
    ncase = 6
    for k in range(ncase):
        dur = rng.uniform(du_min, du_min + (D - 1) * du_dt, P)
        st = rng.uniform(st_min, st_min + (S - 1) * st_dt, (T, P))
        sl = rng.uniform(0, 5, P)
        if k == 1:  # values exactly on grid nodes incl. node 0 (multilinear -1 wrap, A.3)
            dur[:4] = du_min + np.array([0, 1, 2, 3]) * du_dt
            st[:, :5] = st_min + np.array([0, 1, 2, 8, 4]) * st_dt
        if k == 2:  # half-way ties: round-half-even (A.1)
            dur[:3] = du_min + (np.array([0, 1, 2]) + 0.5) * du_dt
            st[:, :6] = st_min + (np.array([0, 1, 2, 3, 4, 7]) + 0.5) * st_dt
        if k == 3:  # slips given as (1,P) (A.6)
            sl = sl.reshape(1, P)
        for interp in ("nearest_neighbor", "multilinear"):
            tag = "c%d_%s" % (k, "nn" if interp == "nearest_neighbor" else "ml")
            o = gfs.stack_all(durations=dur, starttimes=st, slips=sl, targetidxs=tidx,
                              interpolation=interp)
            di, df = gfs.durations2idxs(dur, interpolation=interp)
            si, sf = gfs.starttimes2idxs(st, interpolation=interp)
            out[tag + "_out"] = o
            out[tag + "_di"] = di
            out[tag + "_si"] = si
            if df is not None:
                out[tag + "_df"] = df
                out[tag + "_sf"] = sf
        out["c%d_dur" % k] = dur
        out["c%d_st" % k] = st
        out["c%d_sl" % k] = np.asarray(sl)
    out["ncase"] = np.array(ncase)
    save("stack_all", **out)

    # geodetic
    Pg, Nobs = 30, 47
    gcfg = GeodeticGFLibraryConfig(dimensions=(Pg, Nobs))
    ggf = ffibase.GeodeticGFLibrary(config=gcfg)
    ggf.setup(Pg, Nobs, allocate=True)
    ggf._gfmatrix[:] = rng.standard_normal((Pg, Nobs))
    ggf._stack_switch = {"numpy": ggf._gfmatrix}
    sl = rng.uniform(-1, 3, Pg)
    save("geo_stack", G=ggf._gfmatrix, slips=sl, out=ggf.stack_all(sl))


# ---------------------------------------------------------------- covariance / logp
def gen_cov():
    out = {}
    # reference test/test_covariance.py:71-112 construction (seed 10)
    np.random.seed(10)
    n = 10
    a = np.random.rand(n ** 2).reshape(n, n)
    mats = {"kat": a.T.dot(a) + np.eye(n) * 0.3}
    mats["toeplitz"] = 0.7 ** 2 * rcov.exponential_data_covariance(48, 0.5, 2.0)
    rng = np.random.default_rng(3)
    b = rng.standard_normal((33, 33))
    mats["spd"] = b @ b.T + 33 * np.eye(33)
    mats["ident"] = 0.001 * np.eye(10)  # test_models.py toy data covariance
    for k, Cd in mats.items():
        cov = heart.Covariance(data=Cd)
        out[k + "_C"] = Cd
        out[k + "_W"] = cov.chol_inverse
        out[k + "_inv"] = cov.inverse()
        out[k + "_chol"] = cov.chol()
        out[k + "_logpdet"] = np.array(cov.log_pdet)
        out[k + "_logdet_fn"] = np.array(heart.log_determinant(Cd))
    # data + pred_v sum (c_total, heart.py:158-164)
    cov = heart.Covariance(data=mats["toeplitz"], pred_v=0.1 * np.eye(48))
    out["total_W"] = cov.chol_inverse
    out["total_logpdet"] = np.array(cov.log_pdet)
    out["names"] = np.array(list(mats.keys()))
    out["exp_cov_16"] = rcov.exponential_data_covariance(16, 0.5, 2.0)
    # ensure_cov_psd on an indefinite matrix
    bad = np.array([[1.0, 2.0, 0.3], [2.0, 1.0, 0.1], [0.3, 0.1, 0.5]])
    out["psd_in"] = bad
    out["psd_out"] = utility.ensure_cov_psd(bad)
    save("covariance", **out)


def gen_laplacian():
    out = {}
    for ns, nd, ps, pd in [(7, 5, 1.0, 1.0), (4, 3, 2.0, 2.0), (20, 20, 1.0, 1.0)]:
        L = rlap.get_smoothing_operator_nearest_neighbor(ns, nd, ps, pd)
        tag = "%dx%d" % (nd, ns)
        out[tag + "_L"] = L
        out[tag + "_meta"] = np.array([ns, nd, ps, pd])
        # laplacian.py:57-60: log_determinant(L.T * L) (elementwise product, as-is)
        out[tag + "_logdet"] = np.array(heart.log_determinant(L.T * L, inverse=False))
    save("laplacian", **out)


# ---------------------------------------------------------------- SMC / PT host math
def gen_smc():
    rng = np.random.default_rng(11)
    out = {}
    for k, (n, scale, beta) in enumerate([(100, 50.0, 0.0), (1000, 400.0, 0.01),
                                          (64, 3.0, 0.3), (256, 0.5, 0.7)]):
        lk = rng.standard_normal(n) * scale - 1e3
        ns = SimpleNamespace(beta=beta, coef_variation=1.0, likelihoods=lk, n_chains=n)
        b, ob, w = rsmc.SMC.calc_beta(ns)
        ns.weights = w
        np.random.seed(100 + k)
        aux = np.random.rand(1)
        np.random.seed(100 + k)
        idx = rsmc.SMC.resample(ns)
        pop = rng.standard_normal((n, 6)) * np.array([1, 2, 3, 0.1, 5, 1.0])
        cov = np.cov(pop, aweights=w.ravel(), bias=False, rowvar=0)
        out["c%d_lk" % k] = lk
        out["c%d_beta_in" % k] = np.array(beta)
        out["c%d_beta" % k] = np.array(b)
        out["c%d_w" % k] = w
        out["c%d_aux" % k] = aux
        out["c%d_idx" % k] = idx
        out["c%d_pop" % k] = pop
        out["c%d_cov" % k] = cov
    out["ncase"] = np.array(4)
    acc = np.array([0.0, 0.0005, 0.001, 0.03, 0.05, 0.1, 0.2, 0.3, 0.5, 0.6, 0.75, 0.8, 0.95,
                    0.97, 1.0])
    out["tune_acc"] = acc
    out["pt_tune"] = np.array([rpt.tune(1.2, a) for a in acc])
    out["smc_tune"] = np.array([rsmc.tune(a) for a in acc])
    save("smc", **out)


def gen_noise_cov():
    rng = np.random.default_rng(21)
    out = {}
    for k, (n, win) in enumerate([(64, 8), (97, 11), (200, 20)]):
        t = np.arange(n)
        data = np.sin(t / 7.0) * (1 + 0.5 * np.cos(t / 23.0)) + 0.3 * rng.standard_normal(n)
        out["c%d_data" % k] = data
        out["c%d_win" % k] = np.array(win)
        out["c%d_autocov" % k] = rcov.autocovariance(data)
        out["c%d_rms_same" % k] = utility.running_window_rms(data, win, mode="same")
        out["c%d_rms_valid" % k] = utility.running_window_rms(data, win)
        out["c%d_ntc" % k] = rcov.non_toeplitz_covariance(data, win)
    out["ncase"] = np.array(3)
    save("noise_covariance", **out)


# ---------------------------------------------------------------- Laquila fixture
def gen_laquila():
    path = os.path.join(ref_import.REFERENCE_ROOT, "data/examples/Laquila/geodetic_data.pkl")
    dsets = ref_import.stub_unpickle(path)
    out = {"n": np.array(len(dsets))}
    for i, d in enumerate(dsets):
        dd = d.__dict__
        cv = dd["covariance"].__dict__
        Ct = cv["data"].copy()
        for k in ("pred_g", "pred_v"):
            if cv.get(k) is not None and np.size(cv[k]) == Ct.size:
                Ct = Ct + cv[k]
        cov = heart.Covariance(data=Ct)
        out["d%d_displacement" % i] = np.asarray(dd["displacement"], dtype=np.float64)
        out["d%d_odw" % i] = np.asarray(dd["odw"], dtype=np.float64)
        out["d%d_incidence" % i] = np.asarray(dd["incidence"], dtype=np.float64)
        out["d%d_heading" % i] = np.asarray(dd["heading"], dtype=np.float64)
        out["d%d_C" % i] = Ct
        out["d%d_logpdet" % i] = np.array(cov.log_pdet)
        W = cov.chol_inverse
        out["d%d_W_checksum" % i] = np.array([W.sum(), np.abs(W).sum(), np.trace(W)])
        # LOS through the reference's own expression (heart.py:1381-1398)
        Su = np.cos(np.deg2rad(dd["incidence"]))
        Sn = -np.sin(np.deg2rad(dd["incidence"])) * np.cos(np.deg2rad(dd["heading"] - 270))
        Se = -np.sin(np.deg2rad(dd["incidence"])) * np.sin(np.deg2rad(dd["heading"] - 270))
        out["d%d_los" % i] = np.array([Sn, Se, Su], dtype=np.float64).T
    save("laquila_geodetic", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate selected fixtures only: gen_golden.py sweep_ties ...
        for what in sys.argv[1:]:
            globals()["gen_" + what]()
        sys.exit(0)
    gen_sweep()
    gen_sweep_ties()
    gen_positions()
    gen_stack()
    gen_cov()
    gen_laplacian()
    gen_smc()
    gen_noise_cov()
    gen_laquila()
