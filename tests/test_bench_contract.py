"""bench.py pieces that do not need a GPU: the algorithmic-bytes figure of SURVEY 8(d) and the
cpu_baseline leg (oracle timed on a bounded sample)."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_algorithmic_bytes_match_survey():
    import bench
    from beat_amd.synthetic import SyntheticSpec
    spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, nuc_margin=6.0,
                         time_bounds=(0.0, 0.5))
    b = bench.algorithmic_bytes_per_chain_step(spec)
    assert abs(b - 841.1e6) < 0.2e6  # ~841.1 MB for nearest neighbour
    spec.interpolation = "multilinear"
    assert abs(bench.algorithmic_bytes_per_chain_step(spec) - 3357.6e6) < 2.5e6
    assert spec.lib_bytes == 64 * 400 * 3 * 25 * 4096 * 8  # 62.9 GB


def test_cpu_baseline_leg_small():
    import bench
    from beat_amd.synthetic import SyntheticSpec
    spec = SyntheticSpec((6,), (6,), (1.0,), T=8, N=64, D=3, S=25, nuc_margin=0.0, time_bounds=(0.0, 0.5))
    # the sample draws hypocentres in [6,13) km: widen the toy fault accordingly
    spec = SyntheticSpec((20,), (20,), (1.0,), T=8, N=32, D=3, S=25, nuc_margin=6.0, time_bounds=(0.0, 0.5))
    out = bench.cpu_baseline(spec, seconds=1.0)
    assert set(out) >= {"value", "unit", "cores", "kind", "sample", "value_1core"}
    assert out["kind"] == "port" and out["unit"] == "chain-steps/s" and out["value"] > 0
    assert out["cores"] == len(os.sched_getaffinity(0))


def _stub_result(n_extra_legs=0):
    """a full result object of the shape bench.py builds (last round's committed one), with its verbose notes"""
    import json
    out = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_default.json")))
    for i in range(n_extra_legs):
        out["extra_%d_leg" % i] = {"chain_steps_per_s": 1234.5 + i, "ms_per_step": 3.21, "kernel": "k_gfstack_ws<1,1,3,1>",
                                   "note": "x" * 700, "roofline": {"frac": 0.5, "kernel": "k_gfstack_ws<1,1,3,1>"}}
    return out


def test_bench_line_is_compact_strict_json():
    """the driver keeps an 8 KB tail of stdout and could not take the 40 KB line of round 5 apart; the line
    bench.py prints last must be strict JSON, well below 8 KB, and carry the contract fields + roofline + cpu_baseline"""
    import json

    import bench
    out = _stub_result()
    assert len(json.dumps(out)) > 30000          # (the stub really is the object that broke the driver)
    line = bench.compact_line(out, "/somewhere/bench_full.json")
    assert "\n" not in line and len(line.encode()) < 6144, len(line)
    d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))    # no NaN / Infinity
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "speedup_vs_cpu_baseline", "legs"):
        assert k in d, k
    assert set(d["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                  "algorithmic_bytes_per_launch", "avg_launch_ms"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "value_1core",
                                      "value_1core_numpy_reference_path"}
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["roofline_streaming"]["bound"] == "hbm" and 0 < d["roofline_streaming"]["frac"] < 1
    assert d["full"] == "bench_full.json"
    legs = d["legs"]
    assert {"multilinear", "batch_2048", "toeplitz.banded", "config4.N120.multilinear_512_chains",
            "realistic_grid.nn_512_chains", "smc.2048_chains"} <= set(legs)
    assert all(set(e) <= {"cps", "ms", "k", "frac", "pmc"} for e in legs.values())


def test_bench_line_stays_below_the_cap_whatever_the_legs():
    import json

    import bench
    line = bench.compact_line(_stub_result(n_extra_legs=200), None)
    assert len(line.encode()) < bench.LINE_CAP
    d = json.loads(line)
    assert d.get("legs_truncated") is True and d["roofline"]["kernel"] and d["cpu_baseline"]["value"] > 0
    # NaN in a leg never reaches the line
    out = _stub_result()
    out["multilinear_leg"]["chain_steps_per_s"] = float("nan")
    json.loads(bench.compact_line(out), parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))


def test_committed_round6_line_is_what_the_driver_needs():
    """the line of the round's own `python bench.py` run on the GPU box (profiles/r6_bench_default.json): strict JSON below
    the driver's 8 KB tail, counter-based roofline of the headline kernel, cpu_baseline, the short-trace legs with counter
    fractions"""
    import json
    path = os.path.join(ROOT, "profiles", "r6_bench_default.json")
    if not os.path.exists(path):
        pytest.skip("no committed round-6 line")
    txt = open(path).read().strip().splitlines()[-1]
    assert len(txt.encode()) < 8192
    d = json.loads(txt, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert d["metric"].startswith("SMC chain-steps/s") and d["unit"] == "chain-steps/s" and d["n_gpus"] == 1
    assert abs(d["value"] - 512 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"].startswith("k_gfstack_ws<") and r["traffic"] and 0.5 < r["frac"] < 1.0
    assert abs(r["frac"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    legs = d["legs"]
    for name in ("config4.N120.multilinear_512_chains", "config4.N120.nn_512_chains"):
        assert legs[name].get("pmc") == 1 and legs[name]["frac"] > 0.3, (name, legs[name])
