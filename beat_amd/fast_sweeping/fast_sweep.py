"""Counterpart of beat/fast_sweeping/fast_sweep.py (C-implementation wrapper :24-64)."""
from . import fast_sweep_ext


def get_rupture_times_c(slowness, patch_size, n_patch_strike, n_patch_dip, nuc_x, nuc_y):
    """Same signature as the reference; strike/dip are swapped on purpose when calling the
    extension (fast_sweep.py:55-64): rows of the result run along dip."""
    return fast_sweep_ext.fast_sweep(slowness, patch_size, nuc_y, nuc_x, n_patch_dip, n_patch_strike)


def get_rupture_times_batch(slowness, patch_size, n_patch_strike, n_patch_dip, nuc_x, nuc_y):
    """slowness (C, n_dip*n_strike); nuc_x/nuc_y (C,) int32 -> (C, n)"""
    return fast_sweep_ext.fast_sweep_batch(slowness, patch_size, nuc_y, nuc_x, n_patch_dip,
                                           n_patch_strike)
