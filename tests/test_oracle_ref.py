"""Validate the C restatement against the REFERENCE's own C extension compiled in
place (oracle/_ref, built by oracle/Makefile from /root/reference/beat/fast_sweeping/
fast_sweep_ext.c).  The .so travels with the snapshot; skipped if absent."""
import glob
import importlib.util
import os

import numpy as np
import pytest

from oracle import oracle as orc

_REF = os.path.join(os.path.dirname(os.path.abspath(orc.__file__)), "_ref")


def _load_ref():
    so = glob.glob(os.path.join(_REF, "fast_sweep_ext*.so"))
    if not so:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    spec = importlib.util.spec_from_file_location("fast_sweep_ext", so[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_sweep_bit_exact_random_grids():
    ref = _load_ref()
    rng = np.random.default_rng(42)
    for _ in range(150):
        nd, ns = int(rng.integers(2, 21)), int(rng.integers(2, 21))
        slow = 1.0 / rng.uniform(0.5, 6.0, nd * ns)
        psz = float(rng.uniform(0.5, 3.0))
        hd, hs = int(rng.integers(0, nd)), int(rng.integers(0, ns))
        a = ref.fast_sweep(slow, psz, hd, hs, nd, ns)
        b = orc.fast_sweep(slow, psz, hd, hs, nd, ns)
        assert np.array_equal(a, b)
