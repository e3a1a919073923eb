"""CPU tests: the C-ABI library loads and exports every symbol include/beat_amd.h declares,
fails loudly without a GPU, and the host-side mirrors of the reference interface behave."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    from beat_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "beat_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(beatamd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libbeat_amd.so does not export %s" % name
    # and the python prototype table covers the header
    assert set(declared) == set(_lib.EXPORTS)
    from beat_amd._lib import ABI_VERSION
    assert lib.beatamd_version() == ABI_VERSION == 120


def test_header_cites_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "beat_amd.h")).read()
    for cite in ("fast_sweep_ext.c:120-245", "beat/ffi/base.py:607-709",
                 "beat/models/distributions.py:72-140", "beat/sampler/metropolis.py:276-422",
                 "beat/models/laplacian.py:88-139", "sampler/base.py:598-615"):
        assert cite in hdr


@pytest.mark.skipif(_gpu_present(), reason="GPU present")
def test_no_cpu_fallback_without_gpu():
    import beat_amd
    with pytest.raises(beat_amd.BeatAmdError):
        beat_amd.Context(0)
    from beat_amd.ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    gf = SeismicGFLibrary(SeismicGFLibraryConfig(dimensions=(1, 2, 1, 1, 4)))
    gf.setup(1, 2, 1, 1, 4, allocate=True)
    with pytest.raises(beat_amd.BeatAmdError):
        gf.stack_all(np.zeros(2), np.zeros((1, 2)), np.ones(2), targetidxs=[0])


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under beat_amd/ may reference it"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "beat_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                path = os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), path
                assert "libbeat_oracle" not in txt and "beat_oracle" not in txt, path
                assert "oracle/" not in txt and "oracle." not in txt, path


def test_parameter_layout_and_bounds():
    from beat_amd.models import ParameterLayout, prior_logp_func
    lay = ParameterLayout({"uparr": 3, "durations": 3, "time": 1})
    pt = {"uparr": [1, 2, 3], "durations": [4, 5, 6], "time": [7]}
    q = lay.map(pt)
    assert q.tolist() == [1, 2, 3, 4, 5, 6, 7]
    back = lay.rmap(q)
    assert back["durations"].tolist() == [4, 5, 6] and lay.offset("time") == 6
    with pytest.raises(KeyError):
        lay.offset("velocities")
    lo, up = lay.bounds({"uparr": 0, "durations": 0.5, "time": [-1]}, {"uparr": 5, "durations": 8, "time": [9]})
    f = prior_logp_func(lo, up)
    assert np.isfinite(f(q)) and not np.isfinite(f(q + 100))


def test_gflibrary_host_interface():
    from beat_amd.ffi import (GFLibraryError, SeismicGFLibrary, SeismicGFLibraryConfig)
    g = load_golden("stack_all")
    st_min, st_dt, du_min, du_dt = g["cfg"]
    T, P, D, S, N = g["G"].shape
    gf = SeismicGFLibrary(SeismicGFLibraryConfig(dimensions=g["G"].shape, starttime_sampling=st_dt,
                                                 duration_sampling=du_dt, starttime_min=st_min,
                                                 duration_min=du_min))
    assert (gf.ntargets, gf.npatches, gf.ndurations, gf.nstarttimes, gf.nsamples) == (T, P, D, S, N)
    assert gf.filename == "seismic_uparr_any_P_1_0" and gf.patchidxs.dtype == np.int16
    # index maps are bit-exact twins of the reference (golden from ffi/base.py:486-568)
    for k in range(int(g["ncase"])):
        for interp, tag in (("nearest_neighbor", "nn"), ("multilinear", "ml")):
            di, df = gf.durations2idxs(g["c%d_dur" % k], interp)
            si, sf = gf.starttimes2idxs(g["c%d_st" % k], interp)
            assert np.array_equal(di, g["c%d_%s_di" % (k, tag)]) and di.dtype == np.int16
            assert np.array_equal(si, g["c%d_%s_si" % (k, tag)])
            if tag == "ml":
                assert np.array_equal(sf, g["c%d_ml_sf" % k])
    with pytest.raises(GFLibraryError):
        gf.put(np.zeros((2, 3)), 0, 0, [du_min], [st_min])  # library not allocated
    gf.setup(T, P, D, S, N, allocate=True)
    with pytest.raises(GFLibraryError):
        gf.put(np.zeros((1, N + 1)), 0, 0, [du_min], [st_min])  # wrong trace length
    with pytest.raises(ValueError):
        gf.put(np.zeros(N), 0, 0, [du_min], [st_min])
    durs = du_min + du_dt * np.arange(D)
    sts = st_min + st_dt * np.arange(S)
    gf.put(np.ones((D, N))[:, None, :].repeat(S, 1).reshape(D * S, N)[:D], 0, 0, durs, sts[:D])
    for mode in ("numpy", "pytensor", "hip"):   # the reference's mode names are accepted aliases
        gf.set_stack_mode(mode)
        assert gf.get_stack_mode() == mode
    with pytest.raises(GFLibraryError):
        gf.set_stack_mode("cpu")  # there is no CPU stacking mode
    with pytest.raises(NotImplementedError):
        gf.starttimes2idxs(np.zeros(3), "cubic")


def test_sweeper_and_ext_validation_before_any_gpu_work():
    from beat_amd.fast_sweeping import fast_sweep_ext
    from beat_amd.pytensorf import Sweeper
    from beat_amd.utility import positions2idxs
    a, b = Sweeper(1.0, 6, 4, "c"), Sweeper(1.0, 6, 4, "c")
    assert a == b and hash(a) == hash(b) and a.infer_shape() == [(24,)]
    assert Sweeper.__props__ == ("patch_size", "n_patch_dip", "n_patch_strike", "implementation")
    with pytest.raises(NotImplementedError):
        Sweeper(1.0, 6, 4, "fortran")
    with pytest.raises(AttributeError):
        fast_sweep_ext.fast_sweep([1.0, 2.0], 1.0, 0, 0, 1, 2)
    with pytest.raises(AttributeError):
        fast_sweep_ext.fast_sweep(np.ones(4, dtype=np.float32), 1.0, 0, 0, 2, 2)
    with pytest.raises(AttributeError):
        fast_sweep_ext.fast_sweep(np.ones((4, 4))[:, ::2], 1.0, 0, 0, 4, 2)
    with pytest.raises(fast_sweep_ext.error):
        fast_sweep_ext.fast_sweep(np.ones(4), "x", 0, 0, 2, 2)
    g = load_golden("positions2idxs")
    for cell in (1.0, 2.0, 2.5):
        assert np.array_equal(positions2idxs(g["pos"], cell), g["idx_%g" % cell])


def test_synthetic_problem_shapes_and_oracle_selfconsistency():
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population
    from oracle import problem_oracle
    spec = SyntheticSpec((4, 3), (5, 6), (2.0, 2.0), T=3, N=20, D=3, S=40,
                         slip_varnames=("uparr", "uperp"), covariance="toeplitz",
                         station_shifts=True, geodetic_nobs=(7, 5))
    prob, host = build_problem(spec)
    assert prob.npatches == 38 and prob.out_names[-1] == "like" and len(prob.out_names) == 6
    Q = draw_population(spec, host["layout"], host["lower"], host["upper"], 2)
    ll, ex = problem_oracle.forward(host, Q[0])
    assert ll.shape == (6,) and np.isclose(ll[-1], ll[:-1].sum())
    L = prob.c_layout()
    assert L.nparams == host["layout"].size and L.nvar == 2 and L.h_laplacian_off == -1
    with pytest.raises(ValueError):
        build_problem(SyntheticSpec((20,), (20,), (1.0,), T=2, N=8, D=3, S=5))  # axis too short


def test_chain_block_partition():
    from beat_amd.parallel import chain_block
    for n, w in [(4096, 8), (10, 3), (7, 8), (100, 1)]:
        blocks = [chain_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_isa_audit_of_hidden_asm_loads(tmp_path):
    """k_gfstack_dma hides its slot/weight loads from hipcc; tools/audit_hidden_loads.py checks in
    the generated ISA that their destination registers are not named before the covering wait.
    The detector itself is checked on a synthetic hazard first."""
    import importlib.util
    import os
    import shutil
    spec = importlib.util.spec_from_file_location(
        "audit_hidden_loads", os.path.join(os.path.dirname(__file__), "..", "tools", "audit_hidden_loads.py"))
    aud = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aud)
    gather = ["ds_read_b64 v[40:41], v1 offset:%d" % (8 * i) for i in range(40)]
    loop = [".LBB0_1:", "s_waitcnt vmcnt(0)", "s_barrier", "global_load_ushort v9, v[2:3], off",
            "global_load_dwordx2 v[4:5], v[6:7], off"] + gather + \
           ["v_fma_f64 v[10:11], v[12:13], v[14:15], v[10:11]", "s_cbranch_scc1 .LBB0_1"]
    hidden = lambda body: [ln.startswith("global_load_") for ln in body]   # the asm-statement lines
    ok = loop + ["s_waitcnt vmcnt(0)", "v_mov_b32_e32 v9, v1", "s_endpgm"]
    n, problems = aud.audit(ok, hidden(ok))
    assert n == 2 and not problems
    bad = list(ok)
    bad.insert(0, "v_mov_b32_e32 v20, v9")          # before the loop: not on the path
    bad.insert(2, "v_mov_b64_e32 v[30:31], v[4:5]")  # at the loop header, before the wait
    n, problems = aud.audit(bad, hidden(bad))
    assert len(problems) == 1 and "v[30:31]" in problems[0][3]
    # the way OUT of the loop: the loads of the step after the last are in flight behind the loop
    # (the round-2 fault of k_gfstack_dma: store addresses built in those registers)
    leak = loop + ["v_lshl_add_u64 v[4:5], v[60:61], 0, s[2:3]", "global_store_dwordx2 v[4:5], v[10:11], off",
                   "s_endpgm"]
    n, problems = aud.audit(leak, hidden(leak))
    assert len(problems) == 1 and "v_lshl_add_u64" in problems[0][3]
    # a conditional skip around the only wait is followed too
    skip = loop + ["s_cbranch_execz .LBB0_2", "s_waitcnt vmcnt(0)", ".LBB0_2:", "v_mov_b32_e32 v9, v1", "s_endpgm"]
    n, problems = aud.audit(skip, hidden(skip))
    assert len(problems) == 1 and "v_mov_b32_e32 v9" in problems[0][3]
    # a compiler-generated load (outside asm statements) is not a hidden load
    plain = [".LBB0_1:", "global_load_dwordx2 v[4:5], v[6:7], off", "v_mov_b32_e32 v8, v4", "s_endpgm"]
    assert aud.audit(plain, [False] * len(plain)) == (0, [])
    # a row request in front of the slot/weight loads of its step is flagged (k_gfstack_ws loaders
    # wait with vmcnt(k) > 0)
    deep = [".LBB0_1:", "s_waitcnt vmcnt(3)", "s_barrier", "global_load_lds_dwordx4 v1, s[4:5]",
            "global_load_ushort v9, v[2:3], off"] + gather + ["s_cbranch_scc1 .LBB0_1", "s_endpgm"]
    assert aud.audit_order(deep) == [2]
    deep[3], deep[4] = deep[4], deep[3]
    assert aud.audit_order(deep) == []
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    assert aud.main() == 0


def test_patch_ranges_of_short_trace_libraries():
    """beatamd_gf_patch_ranges (round 6): how many patch ranges a library of short traces is stacked in -- a
    pure function of the library's shape and the device's CU count, no GPU needed.  BASELINE configs[3] at 120 samples: 70
    walks x 400 patches -> 10 ranges (700 walks = 3 rounds of 45 steps on 256 CUs; 8 ranges would be 3 rounds of 55)"""
    from beat_amd import _lib
    lib = _lib.load()
    R = lib.beatamd_gf_patch_ranges
    assert R(35, 400, 120, 256) == 10
    assert R(35, 400, 4096, 256) == 1            # long traces: 35 x 64 tiles are walks enough
    assert R(3, 128, 120, 256) == 4 and R(4, 96, 100, 256) == 3    # (tests/test_gpu_split.py: at least 32 patches per range)
    assert R(600, 400, 120, 256) == 1            # already two walks per CU and more
    assert R(35, 50, 120, 256) == 1              # too few patches to cut
    assert R(35, 400, 120, 0) == R(35, 400, 120, 256)
    # the rule's own cost model: no divisor with >= 32 patches per range is cheaper than the choice
    for T, P, N, cu in ((35, 400, 120, 256), (64, 400, 120, 256), (20, 360, 200, 256), (35, 400, 120, 304), (7, 1024, 64, 256)):
        best = R(T, P, N, cu)
        walks = T * ((N + 63) // 64)
        cost = lambda d: -(-walks * d // cu) * (P // d + 5)
        cands = [d for d in range(1, 33) if P % d == 0 and (d == 1 or P // d >= 32)]
        assert best in cands and all(cost(best) <= cost(d) for d in cands), (T, P, N, cu, best)
