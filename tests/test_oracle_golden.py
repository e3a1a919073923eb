"""Pin the CPU oracle (oracle/) against golden vectors produced by the REFERENCE's
own numpy path and C extension (oracle/gen_golden.py, run where /root/reference
exists).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc


def test_sweep_bit_exact_vs_reference_c(golden):
    g = golden("sweep")
    for name in g["names"]:
        slow = g[name + "_slow"]
        psz, hd, hs, nd, ns = g[name + "_meta"]
        out = orc.fast_sweep(slow.ravel(), psz, int(hd), int(hs), int(nd), int(ns))
        # the restatement executes the same double operations as fast_sweep_ext.c
        assert np.array_equal(out, g[name + "_c"]), name
        # the reference's numpy twin is only ulp-close to its own C
        np.testing.assert_allclose(out, g[name + "_numpy"], rtol=0, atol=1e-6)


def test_positions2idxs(golden):
    g = golden("positions2idxs")
    for cell in (1.0, 2.0, 2.5):
        assert np.array_equal(orc.positions2idxs(g["pos"], cell), g["idx_%g" % cell])


@pytest.mark.parametrize("interp,tag", [("nearest_neighbor", "nn"), ("multilinear", "ml")])
def test_stack_all(golden, interp, tag):
    g = golden("stack_all")
    st_min, st_dt, du_min, du_dt = g["cfg"]
    G = g["G"]
    for k in range(int(g["ncase"])):
        dur, st, sl = g["c%d_dur" % k], g["c%d_st" % k], g["c%d_sl" % k]
        di, df = orc.time2idx(dur, du_min, du_dt, interp)
        si, sf = orc.time2idx(st, st_min, st_dt, interp)
        pre = "c%d_%s" % (k, tag)
        assert np.array_equal(di, g[pre + "_di"])  # int16, bit exact
        assert np.array_equal(si, g[pre + "_si"])
        if interp == "multilinear":
            assert np.array_equal(df, g[pre + "_df"])
            assert np.array_equal(sf, g[pre + "_sf"])
        out = orc.stack_all(G, dur, st, sl, du_min, du_dt, st_min, st_dt, interp)
        np.testing.assert_allclose(out, g[pre + "_out"], rtol=1e-12, atol=1e-12)


def test_geo_stack(golden):
    g = golden("geo_stack")
    np.testing.assert_allclose(orc.geo_stack(g["G"], g["slips"]), g["out"], rtol=1e-13,
                               atol=1e-13)


def test_covariance(golden):
    g = golden("covariance")
    for k in g["names"]:
        C = g[k + "_C"]
        np.testing.assert_allclose(orc.cov_chol_inverse(C), g[k + "_W"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(orc.cov_log_pdet(C), g[k + "_logpdet"], rtol=1e-13)
        np.testing.assert_allclose(orc.log_determinant(C), g[k + "_logdet_fn"], rtol=1e-13)
    np.testing.assert_array_equal(orc.exponential_data_covariance(16, 0.5, 2.0),
                                  g["exp_cov_16"])
    Ct = g["toeplitz_C"] + 0.1 * np.eye(48)
    np.testing.assert_allclose(orc.cov_chol_inverse(Ct), g["total_W"], rtol=1e-10, atol=1e-12)


def test_laplacian(golden):
    g = golden("laplacian")
    for tag in ("5x7", "3x4", "20x20"):
        ns, nd, ps, pd = g[tag + "_meta"]
        L = orc.smoothing_operator_nearest_neighbor(int(ns), int(nd), ps, pd)
        assert np.array_equal(L, g[tag + "_L"])
        np.testing.assert_allclose(orc.log_determinant(L.T * L), g[tag + "_logdet"], rtol=1e-13)
        # reference test/test_laplacian.py:28-52: operator rows sum to zero
        np.testing.assert_allclose(L.sum(1), 0.0, atol=1e-12)


def test_smc_host_math(golden):
    g = golden("smc")
    for k in range(int(g["ncase"])):
        lk = g["c%d_lk" % k]
        b, ob, w = orc.calc_beta(lk, float(g["c%d_beta_in" % k]))
        np.testing.assert_allclose(b, g["c%d_beta" % k], rtol=0, atol=2e-6)
        if abs(b - float(g["c%d_beta" % k])) == 0.0:
            np.testing.assert_allclose(w, g["c%d_w" % k], rtol=1e-10)
        idx = orc.resample(g["c%d_w" % k], float(g["c%d_aux" % k][0]))
        assert np.array_equal(idx, g["c%d_idx" % k])
        cov = orc.weighted_covariance(g["c%d_pop" % k], g["c%d_w" % k])
        np.testing.assert_allclose(cov, g["c%d_cov" % k], rtol=1e-9, atol=1e-12)
    for a, pt_s, smc_s in zip(g["tune_acc"], g["pt_tune"], g["smc_tune"]):
        assert orc.pt_tune(1.2, a) == pt_s
        assert orc.smc_tune(a) == smc_s


def test_laquila_fixture(golden):
    g = golden("laquila_geodetic")
    for i in range(int(g["n"])):
        C = g["d%d_C" % i]
        W = orc.cov_chol_inverse(C)
        chk = np.array([W.sum(), np.abs(W).sum(), np.trace(W)])
        np.testing.assert_allclose(chk, g["d%d_W_checksum" % i], rtol=1e-9)
        np.testing.assert_allclose(orc.cov_log_pdet(C), g["d%d_logpdet" % i], rtol=1e-12)
        los = orc.los_vectors(g["d%d_incidence" % i], g["d%d_heading" % i])
        assert np.array_equal(los, g["d%d_los" % i])


def test_noise_covariance(golden):
    g = golden("noise_covariance")
    for k in range(int(g["ncase"])):
        d, w = g["c%d_data" % k], int(g["c%d_win" % k])
        assert np.array_equal(orc.autocovariance(d), g["c%d_autocov" % k])
        assert np.array_equal(orc.running_window_rms(d, w, "same"), g["c%d_rms_same" % k])
        assert np.array_equal(orc.non_toeplitz_covariance(d, w), g["c%d_ntc" % k])


def test_sweep_tie_fixtures_oracle_bit_exact():
    """rupture times that sit exactly on k + 0.5 ties of the start-time grid
    (tests/golden/sweep_ties.npz, times from the reference's C extension): the C restatement
    reproduces them bit for bit and maps them to the reference's int16 indices (round-half-even)"""
    g = load_golden("sweep_ties")
    n_ties = 0
    for name in g["names"]:
        slow = g[name + "_slow"]
        psz, hd, hs, nd, ns = g[name + "_meta"]
        out = orc.fast_sweep(slow.ravel(), psz, int(hd), int(hs), int(nd), int(ns))
        assert np.array_equal(out, g[name + "_c"]), name
        for dt in (0.5, 0.25):
            tie = g[name + "_ties_%g" % dt]
            n_ties += int(tie.sum())
            assert np.array_equal(orc.time2idx(out, 0.0, dt)[0], g[name + "_idx_nn_%g" % dt])
            assert np.array_equal(orc.time2idx(out, 0.0, dt, "multilinear")[0], g[name + "_idx_ml_%g" % dt])
            # half-even: a tie k + 0.5 goes to the even neighbour
            k = np.floor(out[tie] / dt)
            want = np.where(k % 2 == 0, k, k + 1)
            assert np.array_equal(orc.time2idx(out, 0.0, dt)[0][tie], want.astype(np.int16))
    assert n_ties >= 50
