"""bench.py pieces that do not need a GPU: the algorithmic-bytes figure of SURVEY 8(d) and the
cpu_baseline leg (oracle timed on a bounded sample)."""
import os
import sys

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
