"""
Drop-in for the reference's native module ``fast_sweep_ext``
(beat/fast_sweeping/fast_sweep_ext.c:208-245): same function name, argument order,
array validation and exception types -- the sweep itself runs on the GPU.
"""
import numpy as np

from ..engine import get_context


class error(Exception):  # beat.fast_sweep_ext.error (fast_sweep_ext.c:286)
    pass


def _good_array(o):
    """fast_sweep_ext.c:18-56 good_array: failures raise AttributeError"""
    if not isinstance(o, np.ndarray):
        raise AttributeError("not a NumPy array")
    if o.dtype != np.float64:
        raise AttributeError("array of unexpected type")
    if not (o.flags.c_contiguous and o.flags.aligned and o.flags.writeable):
        raise AttributeError("array is not contiguous or not well behaved")


def fast_sweep(slowness_arr, patch_size, h_strk, h_dip, num_strk, num_dip):
    """fast_sweep(slowness_arr, patch_size, h_strk, h_dip, num_strk, num_dip) -> float64[n]"""
    try:
        patch_size = float(patch_size)
        h_strk, h_dip, num_strk, num_dip = int(h_strk), int(h_dip), int(num_strk), int(num_dip)
    except (TypeError, ValueError):
        raise error("Invalid call to fast_sweep! \n usage: fast_sweep(slowness_arr, patch_size, "
                    "h_strk, h_dip, num_strk, num_dip)")
    _good_array(slowness_arr)
    out = get_context().fast_sweep_batch(slowness_arr.reshape(1, -1), patch_size, [h_strk], [h_dip],
                                         num_strk, num_dip)
    return out[0]


def fast_sweep_batch(slowness, patch_size, h_strk, h_dip, num_strk, num_dip):
    """Batched form: slowness (C, n), h_strk/h_dip (C,) -> (C, n); numpy or torch-cuda."""
    return get_context().fast_sweep_batch(slowness, patch_size, h_strk, h_dip, num_strk, num_dip)
