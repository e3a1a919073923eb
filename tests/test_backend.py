"""Trace writers against the reference's own fixtures test/PT_TEST/chain-0.{bin,csv}
(copied as data into tests/golden/): byte-identical output.  CPU only."""
import os
from collections import OrderedDict

import numpy as np

from beat_amd.backend import NumpyChain, TextChain, flat_names_of, stage_path, write_population
from conftest import GOLDEN

SHAPES = OrderedDict([("Data", (5,)), ("A", (5,)), ("F", (5,)), ("D", (5,)), ("B", (5,)),
                      ("Hood", (5,)), ("like", ())])


def _lpoint():
    """reference test/test_backend.py sample: arange(5) * k per variable, like = 10"""
    return [np.arange(5, dtype=float) * k for k in range(1, 7)] + [np.array(10.0)]


def test_numpychain_bytes_match_reference_fixture(tmp_path):
    ch = NumpyChain(str(tmp_path), SHAPES)
    ch.setup(5, 0, overwrite=True)
    for i in range(5):
        ch.buffer_write(_lpoint(), i)
    ch.record_buffer()
    ref = open(os.path.join(GOLDEN, "PT_TEST_chain-0.bin"), "rb").read()
    assert open(ch.filename, "rb").read() == ref
    # and it reads the reference's file
    r = NumpyChain.load(os.path.join(GOLDEN, "PT_TEST_chain-0.bin"))
    assert r.varnames == list(SHAPES) and len(r) == 5
    np.testing.assert_array_equal(r.get_values("F"), np.tile(np.arange(5) * 3.0, (5, 1)))
    assert r.get_values("like").shape == (5,) and r.point(4)["like"] == 10.0
    assert flat_names_of("A", (2, 2)) == ["A__0_0", "A__0_1", "A__1_0", "A__1_1"]


def test_textchain_bytes_match_reference_fixture(tmp_path):
    ch = TextChain(str(tmp_path), SHAPES)
    ch.setup(5, 0, overwrite=True)
    for i in range(5):
        ch.write(_lpoint())
    ref = open(os.path.join(GOLDEN, "PT_TEST_chain-0.csv")).read()
    assert open(ch.filename).read() == ref
    np.testing.assert_array_equal(ch.get_values("Hood")[2], np.arange(5) * 6.0)


def test_write_population_roundtrip(tmp_path):
    from beat_amd.models import ParameterLayout
    lay = ParameterLayout(OrderedDict([("uparr", 4), ("durations", 4), ("h_any_P_0_Z", 1)]))
    names = ["seis_like_any_P_0_0", "seis_like_any_P_0_1", "geo_like_0", "laplacian_like", "like"]
    rng = np.random.default_rng(0)
    pop, lp = rng.random((3, 9)), rng.random((3, 5))
    path = write_population(str(tmp_path), 2, lay, names, pop, lp)
    assert path == stage_path(str(tmp_path), 2) and path.endswith("stage_2")
    assert stage_path("x", -1).endswith("stage_final")
    ch = NumpyChain.load(os.path.join(path, "chain-1.bin"))
    np.testing.assert_array_equal(ch.get_values("durations")[0], pop[1, 4:8])
    np.testing.assert_array_equal(ch.get_values("seis_like")[0], lp[1, :2])
    assert ch.get_values("like")[0] == lp[1, 4] and ch.get_values("geo_like").shape == (1, 1)
