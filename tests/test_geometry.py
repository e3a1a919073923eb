"""Geometry-mode geodetic composite (BASELINE configs 1 and 2): the analytic half-space
sources.  Parity with BEAT/pyrocko is UNPINNED here (layered GF stores, arithmetic not in the
reference tree); the pins are Okada's (1985) published check values and closed forms."""
from collections import OrderedDict

import numpy as np
import pytest

from conftest import load_golden
from oracle import okada_oracle as ok
from oracle import oracle as orc


def test_okada85_published_check_values():
    """Okada (1985) Table 2, case 2 (dip 70 deg) and case 3 (vertical fault)"""
    want = {(1, 0, 0): (-8.689e-3, -4.298e-3, -2.747e-3),
            (0, 1, 0): (-4.682e-3, -3.527e-2, -3.564e-2),
            (0, 0, 1): (-2.660e-4, 1.056e-2, 3.214e-3)}
    for U, ref in want.items():
        got = ok.okada85_local(2.0, 3.0, 4.0, 70.0, 3.0, 2.0, *U)
        np.testing.assert_allclose(got, ref, rtol=6e-4)
    want90 = {(1, 0, 0): (0.0, 5.253e-3, 0.0), (0, 1, 0): (0.0, 0.0, 0.0),
              (0, 0, 1): (1.223e-2, 0.0, -1.606e-2)}
    for U, ref in want90.items():
        got = ok.okada85_local(0.0, 0.0, 4.0, 90.0, 3.0, 2.0, *U)
        np.testing.assert_allclose(got, ref, rtol=6e-4, atol=1e-12)


def test_rect_source_symmetries_and_mogi():
    e, n = np.meshgrid(np.linspace(-9, 9, 7), np.linspace(-9, 9, 7))
    # vertical strike-slip fault striking north through the origin: un odd in e, uz = 0 on n = 0
    ue, un, uz = ok.rect_source(e, n, 0, 0, 1.0, 0.0, 90.0, 0.0, 6.0, 4.0, 1.0)
    np.testing.assert_allclose(un[3], -un[3, ::-1], atol=1e-14)
    np.testing.assert_allclose(uz[3], 0.0, atol=1e-14)
    # rotating the source rotates the field
    ue2, un2, uz2 = ok.rect_source(n, -e, 0, 0, 1.0, 90.0, 90.0, 0.0, 6.0, 4.0, 1.0)
    np.testing.assert_allclose(uz2, uz, atol=1e-14)
    # a small horizontal tensile square far away looks like a Mogi-type vertical pattern: uz > 0 above
    _, _, uzs = ok.rect_source(np.zeros(1), np.zeros(1), 0, 0, 3.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0)
    assert uzs[0] > 0
    # Mogi closed form
    ue, un, uz = ok.mogi(np.array([0.0, 2.0]), np.array([0.0, 0.0]), 0, 0, 2.0, 1e6)
    np.testing.assert_allclose(uz[0], 0.75 / np.pi * 1e6 / 2000.0 ** 2)
    np.testing.assert_allclose(ue[1] / uz[1], 1.0)


def _problem(rng, sizes, two_sources=False):
    from beat_amd.models import GeodeticGeometryProblem, ParameterLayout, los_vectors
    g = load_golden("laquila_geodetic")
    nobs = sum(sizes)
    east, north = rng.uniform(-15, 15, nobs), rng.uniform(-15, 15, nobs)
    inc = np.concatenate([g["d%d_incidence" % i][:n] for i, n in enumerate(sizes)])
    head = np.concatenate([g["d%d_heading" % i][:n] for i, n in enumerate(sizes)])
    los = los_vectors(inc, head)
    ns = 2 if two_sources else 1
    names = OrderedDict([("depth", ns), ("dip", ns), ("east_shift", ns), ("length", 1 if two_sources else ns),
                         ("north_shift", ns), ("slip", ns), ("strike", ns), ("width", 1 if two_sources else ns),
                         ("h_SAR", 1)])
    lay = ParameterLayout(names)
    lower = dict(depth=0.5, dip=5.0, east_shift=-5.0, length=0.5, north_shift=-5.0, slip=0.01, strike=0.0,
                 width=0.5, h_SAR=-2.0)
    upper = dict(depth=9.0, dip=85.0, east_shift=5.0, length=10.0, north_shift=5.0, slip=1.0, strike=360.0,
                 width=8.0, h_SAR=2.0)
    data = 0.01 * rng.standard_normal(nobs)
    odw = 0.5 + rng.random(nobs)
    Ws, sls, o = [], [], 0
    for i, n in enumerate(sizes):
        C = g["d%d_C" % i][:n, :n]
        Ws.append(orc.cov_chol_inverse(C))
        sls.append(orc.cov_log_pdet(C))
    sources = ["rectangular", "mogi"] if two_sources else ["rectangular"]
    fixed = dict(rake=[0.0, 0.0], opening_fraction=[1.0, 0.0]) if two_sources else \
        dict(rake=30.0, opening_fraction=0.25)
    prob = GeodeticGeometryProblem(lay, sources, east, north, los, data, odw, sizes, Ws, sls,
                                   [("h_SAR", 0)] * len(sizes), fixed=fixed, lower=lower, upper=upper)
    return prob, lay, lower, upper


def _oracle_forward(prob, lay, q):
    pt = lay.rmap(q)
    mu = np.zeros(prob.east.size)
    for s, kind in enumerate(prob.sources):
        def val(name):
            if name in lay.offsets:
                return pt[name][s if lay.varsizes[name] > 1 else 0]
            return np.atleast_1d(prob.fixed.get(name, 0.0))[min(s, np.size(prob.fixed.get(name, 0.0)) - 1)]
        if kind == "mogi":
            ue, un, uz = ok.mogi(prob.east, prob.north, val("east_shift"), val("north_shift"), val("depth"),
                                 val("slip"), prob.nu)
        else:
            ue, un, uz = ok.rect_source(prob.east, prob.north, val("east_shift"), val("north_shift"),
                                        val("depth"), val("strike"), val("dip"), val("rake"), val("length"),
                                        val("width"), val("slip"), val("opening_fraction"), prob.nu)
        mu += (un * prob.los[:, 0] + ue * prob.los[:, 1]) + uz * prob.los[:, 2]
    res = (prob.data - mu) * prob.odws
    out, o = [], 0
    for n, W, sl in zip(prob.sizes, prob.weights, prob.slog_pdets):
        out.append(orc.mvn_chol_logp(W, res[o:o + n], sl, pt["h_SAR"][0]))
        o += n
    return np.array(out + [sum(out)])


def test_source_tables_host_logic():
    prob, lay, lower, upper = _problem(np.random.default_rng(0), (20, 10), two_sources=True)
    off, fix = prob.source_tables()
    assert off.shape == (2, 10) and off[0, 2] == lay.offset("depth", 0) and off[1, 2] == lay.offset("depth", 1)
    assert off[0, 6] == off[1, 6] == lay.offset("length", 0)  # shared scalar variable
    assert off[0, 9] == -1 and fix[0, 9] == 1.0 and fix[1, 9] == 0.0
    assert prob.out_names == ["geo_like_0", "geo_like_1", "like"]


@pytest.mark.gpu
@pytest.mark.parametrize("two", [False, True])
def test_geometry_logp_batch_vs_oracle(two):
    import beat_amd
    ctx = beat_amd.get_context(0)
    rng = np.random.default_rng(5 + two)
    prob, lay, lower, upper = _problem(rng, (214, 205) if not two else (60, 41), two_sources=two)
    f = prob.compile(ctx)
    lo, up = lay.bounds(lower, upper)
    C = 70
    Q = lo + (up - lo) * rng.random((C, lay.size))
    if two:
        Q[:, lay.offset("slip", 1)] *= 1e6  # Mogi volume change [m^3]
    LL = f.batch(Q)
    assert LL.shape == (C, len(prob.sizes) + 1)
    for c in range(0, C, 9):
        ref = _oracle_forward(prob, lay, Q[c])
        np.testing.assert_allclose(LL[c], ref, rtol=1e-6)   # north_star tolerance
        np.testing.assert_allclose(LL[c], ref, rtol=1e-9)


@pytest.mark.gpu
def test_geometry_config2_smc_1024_chains():
    """BASELINE configs[1]: rectangular-source geodetic composite, 1024 SMC chains batched on one
    GPU (Laquila observation geometry, Fernandina-style priors)"""
    import torch

    import beat_amd
    from beat_amd.sampler import SMC, smc_sample
    ctx = beat_amd.get_context(0)
    rng = np.random.default_rng(11)
    prob, lay, lower, upper = _problem(rng, (214, 205))
    lo, up = lay.bounds(lower, upper)
    truth = lo + (up - lo) * rng.random(lay.size)
    truth[lay.offset("h_SAR")] = 0.0
    mu_ll = _oracle_forward(prob, lay, truth)  # noqa: F841  (exercise the oracle at the truth)
    f = prob.compile(ctx)
    step = SMC(f, lo, up, n_chains=1024, tune_interval=10, device=torch.device("cuda", 0), random_seed=2)
    pop, lp, betas = smc_sample(15, step, max_stages=6)
    assert pop.shape == (1024, lay.size) and np.isfinite(lp).all()
    np.testing.assert_allclose(f.batch(np.ascontiguousarray(pop[:64])), lp[:64], rtol=1e-12)
    assert lp[:, -1].mean() > f.batch(lo + (up - lo) * rng.random((256, lay.size)))[:, -1].mean()


@pytest.mark.gpu
def test_geometry_special_points_match_oracle():
    """observation points on the fault trace, above the corners and on the strike line; vertical
    and nearly horizontal faults; a surface-breaking fault: kernel and oracle agree, including
    where Okada's expressions are singular (both NaN/inf or both finite, never one of each)"""
    import beat_amd
    from beat_amd.models import GeodeticGeometryProblem, ParameterLayout
    ctx = beat_amd.get_context(0)
    xs = np.array([-6.0, -3.0, -1.5, 0.0, 1.5, 3.0, 6.0])
    east, north = [a.ravel() for a in np.meshgrid(xs, xs)]
    nobs = east.size
    los = np.tile(np.array([[0.3, -0.5, 0.81]]), (nobs, 1))
    lay = ParameterLayout(OrderedDict([("depth", 1), ("dip", 1), ("strike", 1), ("slip", 1), ("h_SAR", 1)]))
    prob = GeodeticGeometryProblem(lay, ["rectangular"], east, north, los, np.zeros(nobs), np.ones(nobs), (nobs,),
                                   [1.0], [0.0], [("h_SAR", 0)],
                                   fixed=dict(east_shift=0.0, north_shift=0.0, rake=45.0, length=6.0, width=3.0,
                                              opening_fraction=0.2))
    f = prob.compile(ctx)
    cases = np.array([[1.0, 90.0, 0.0, 1.0, 0.0],      # vertical, striking north through grid points
                      [0.0, 45.0, 90.0, 1.0, 0.0],     # surface breaking, striking east
                      [2.0, 1e-3, 30.0, 0.5, 0.0],     # nearly horizontal
                      [1.5, 89.999, 180.0, 2.0, 0.5],
                      [0.5, 30.0, 270.0, 1.0, -0.5]])
    LL = f.batch(cases)
    for i, q in enumerate(cases):
        with np.errstate(all="ignore"):
            ref = _oracle_forward(prob, lay, q)
        both_bad = ~np.isfinite(ref) & ~np.isfinite(LL[i])
        ok = np.isfinite(ref) & np.isfinite(LL[i])
        assert (both_bad | ok).all(), (i, ref, LL[i])
        np.testing.assert_allclose(LL[i][ok], ref[ok], rtol=1e-8)


# ------------------------------------------------------------------ heart.geo_synthetics seam
def _seam_case():
    from beat_amd.heart import HalfspaceSource, StaticTarget
    rng = np.random.default_rng(21)
    t1 = StaticTarget(rng.uniform(-20e3, 20e3, 17), rng.uniform(-20e3, 20e3, 17))
    t2 = StaticTarget(rng.uniform(-20e3, 20e3, 5), rng.uniform(-20e3, 20e3, 5))
    s1 = HalfspaceSource("rectangular", east_shift=1500.0, north_shift=-2500.0, depth=3000.0, strike=37.0,
                         dip=62.0, rake=-70.0, length=9000.0, width=5000.0, slip=1.3, opening_fraction=0.2)
    s2 = HalfspaceSource("mogi", east_shift=-4000.0, north_shift=3000.0, depth=4500.0, volume_change=2.0e6)
    return [t1, t2], [s1, s2]


def _seam_reference(targets, sources):
    from oracle import okada_oracle as ok
    out = []
    for s in sources:
        for t in targets:
            e, n = t.east_shifts / 1e3, t.north_shifts / 1e3
            if s.kind == "mogi":
                ue, un, uz = ok.mogi(e, n, s.east_shift / 1e3, s.north_shift / 1e3, s.depth / 1e3, s.volume_change)
            else:
                ue, un, uz = ok.rect_source(e, n, s.east_shift / 1e3, s.north_shift / 1e3, s.depth / 1e3,
                                            s.strike, s.dip, s.rake, s.length / 1e3, s.width / 1e3, s.slip,
                                            s.opening_fraction)
            out.append(np.vstack([un, ue, uz]).T)
    return out


@pytest.mark.gpu
def test_geo_synthetics_seam_outmodes_vs_oracle():
    """heart.geo_synthetics(engine, targets, sources, outmode) (heart.py:4158-4239): per-(source,
    target) [n, e, up] arrays in the reference's order and its four output modes, half-space engine
    against the CPU oracle"""
    from beat_amd.heart import HalfspaceEngine, geo_synthetics
    targets, sources = _seam_case()
    eng = HalfspaceEngine(nu=0.25)
    ref = _seam_reference(targets, sources)
    arrays = geo_synthetics(eng, targets, sources, outmode="arrays")
    assert len(arrays) == 4 and [a.shape for a in arrays] == [(17, 3), (5, 3), (17, 3), (5, 3)]
    for a, r in zip(arrays, ref):
        np.testing.assert_allclose(a, r, rtol=1e-9, atol=1e-14)
    np.testing.assert_array_equal(geo_synthetics(eng, targets, sources, outmode="array"), np.vstack(arrays))
    st = geo_synthetics(eng, targets, sources, outmode="stacked_arrays")
    np.testing.assert_allclose(st[0], ref[0] + ref[2], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(st[1], ref[1] + ref[3], rtol=1e-9, atol=1e-14)
    sa = geo_synthetics(eng, targets, sources)           # default "stacked_array"
    assert sa.shape == (22, 3)
    np.testing.assert_array_equal(sa, np.vstack(st))
    with pytest.raises(ValueError):
        geo_synthetics(eng, targets, sources, outmode="traces")
    with pytest.raises(TypeError):
        geo_synthetics(object(), targets, sources)


@pytest.mark.gpu
def test_geo_synthesizer_op_protocol():
    """pytensorf.GeoSynthesizer (pytensorf.py:25-126): dict inputs in km, mapping to sources,
    perform() writes output[0][0] of shape infer_shape()"""
    import pickle

    from beat_amd.heart import HalfspaceEngine, geo_synthetics
    from beat_amd.pytensorf import GeoSynthesizer
    targets, sources = _seam_case()
    op = GeoSynthesizer(HalfspaceEngine(), sources, targets, mapping={"depth": [0, 1], "slip": [0]})
    assert GeoSynthesizer.__props__ == ("engine", "sources", "targets", "mapping")
    assert op.infer_shape() == [(22, 3)]
    out = op({"depth": np.array([2.0, 6.0]), "slip": np.array([0.7])})    # km, m
    assert sources[0].depth == 2000.0 and sources[1].depth == 6000.0 and sources[0].slip == 0.7
    np.testing.assert_array_equal(out, geo_synthetics(op.engine, targets, sources))
    o2 = [[None]]
    op.make_node({"depth": None, "slip": None})
    op.perform(None, [np.array([2.0, 6.0]), np.array([0.7])], o2)
    np.testing.assert_array_equal(o2[0][0], out)
    op2 = pickle.loads(pickle.dumps(op))
    assert op2.varnames == ["depth", "slip"] and op2.nobs == 22


def test_seis_synthetics_seam_stacks_sources_like_the_reference():
    """heart.seis_synthetics (heart.py:3564-3762): no waveform engine ships with the package; the
    seam stacks an engine's post-processed traces over the sources (:3719-3724)"""
    from beat_amd.heart import seis_synthetics
    from beat_amd.pytensorf import SeisSynthesizer

    class Eng(object):
        def seismograms(self, sources, targets, **kw):
            ns, nt = len(sources), len(targets)
            return np.arange(ns * nt * 4, dtype=float).reshape(ns * nt, 4), np.arange(nt) * 0.5

    srcs, tgts = [object(), object(), object()], [object(), object()]
    out, tmins = seis_synthetics(Eng(), srcs, tgts, outmode="array")
    full = np.arange(24, dtype=float).reshape(6, 4)
    np.testing.assert_array_equal(out, full[0:2] + full[2:4] + full[4:6])
    np.testing.assert_array_equal(tmins, [0.0, 0.5])
    with pytest.raises(NotImplementedError):
        seis_synthetics(object(), srcs, tgts)
    with pytest.raises(TypeError):
        seis_synthetics(Eng(), srcs, tgts, outmode="spectrum")
    assert SeisSynthesizer.__props__[0] == "engine" and len(SeisSynthesizer.__props__) == 13


@pytest.mark.gpu
def test_geometry_stage_from_a_graph_equals_the_eager_loop():
    """BASELINE configs[1] (rectangular source, two SAR scenes with full
    covariances, 1024 chains) is launch-bound -- the Metropolis step of a stage is captured in a HIP
    graph and replayed with the Philox step counter on the device; populations, likelihoods and
    acceptance equal the eager loop bit for bit"""
    import time
    import torch
    import beat_amd
    from beat_amd.sampler import SMC
    from beat_amd.synthetic import build_geometry_problem
    ctx = beat_amd.get_context(0)
    prob, lay, lower, upper = build_geometry_problem()
    lo, up = lay.bounds(lower, upper)
    f = prob.compile(ctx)
    dev = torch.device("cuda", 0)
    out = {}
    for use_graph in (False, True):
        step = SMC(f, lo, up, n_chains=1024, tune_interval=7, device=dev, random_seed=2, use_graph=use_graph)
        Q = step.initialize_population()
        L = step.stepper.evaluate(Q)
        step.select_end_points(Q, L)
        for stage in range(2):
            step.transition()
            step.stage += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Q, L = step.sample_stage(60)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            step.select_end_points(Q, L)
        out[use_graph] = (Q.cpu().numpy(), L.cpu().numpy(), list(step.stage_acceptance),
                          step.stepper.scaling.cpu().numpy(), dt / 60 * 1e6)
    (Qa, La, acca, sca, us_eager), (Qb, Lb, accb, scb, us_graph) = out[False], out[True]
    assert np.array_equal(Qa, Qb) and np.array_equal(La, Lb) and acca == accb and np.array_equal(sca, scb)
    assert np.isfinite(La[:, -1]).all() and 0.0 < acca[-1] < 1.0
    print("geometry stage, 1024 chains: %.1f us per step eager, %.1f us from the graph" % (us_eager, us_graph))
