"""
ctypes binding of ``libbeat_amd.so`` (C ABI declared in ``include/beat_amd.h``).

There is no CPU fallback: if the HIP library is missing, or no GPU is present when a
context is created, the error is raised to the caller.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BEATAMD_LIB: another build of the library (kernel A/B experiments in one GPU session)
LIB_PATH = os.environ.get("BEATAMD_LIB") or os.path.join(_HERE, "libbeat_amd.so")

NEAREST_NEIGHBOR, MULTILINEAR = 0, 1
W_SCALAR, W_DENSE = 0, 1
INTERPOLATIONS = {"nearest_neighbor": NEAREST_NEIGHBOR, "multilinear": MULTILINEAR}

OK, EINVAL, EHIP, EINDEX, ENOMEM, ENAN, ENOTPSD, EBADCOV = 0, -1, -2, -3, -4, -5, -6, -7
ABI_VERSION = 120   # include/beat_amd.h BEATAMD_VERSION this module was written against


class BeatAmdError(RuntimeError):
    pass


class FfiLayout(C.Structure):
    """mirror of ``beatamd_ffi_layout``"""
    _fields_ = [
        ("nparams", C.c_int64),
        ("nvar", C.c_int32),
        ("slip_off", C.c_int64 * 4),
        ("durations_off", C.c_int64),
        ("velocities_off", C.c_int64),
        ("nuc_strike_off", C.c_int64),
        ("nuc_dip_off", C.c_int64),
        ("time_off", C.c_int64),
        ("h_laplacian_off", C.c_int64),
    ]


_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_pi32, _pi64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

# name -> argtypes.  Array arguments are void* so that both numpy (host) and device
# pointers (ints) can be passed.
_PROTOS = {
    "beatamd_ctx_create": [C.c_int, C.POINTER(_vp)],
    "beatamd_ctx_destroy": [_vp],
    "beatamd_ctx_set_stream": [_vp, _vp],
    "beatamd_ctx_use_own_stream": [_vp],
    "beatamd_ctx_synchronize": [_vp],
    "beatamd_ctx_set_step_counter": [_vp, _vp],
    "beatamd_ctx_enable_timing": [_vp, C.c_int],
    "beatamd_ctx_kernel_time": [_vp, C.c_char_p, C.POINTER(_f64), _pi64],
    "beatamd_ctx_reset_timing": [_vp],
    "beatamd_fast_sweep_batch": [_vp, _vp, _f64, _vp, _vp, _i32, _i32, _i64, _vp],
    "beatamd_seis_gflib_create": [_vp, _i64, _i64, _i64, _i64, _i64, _f64, _f64, _f64, _f64, _pi32],
    "beatamd_seis_gflib_upload": [_vp, _i32, _vp, _i64, _i64],
    "beatamd_seis_gflib_adopt": [_vp, _i32, _vp],
    "beatamd_seis_gflib_round_to_f32": [_vp, _i32],
    "beatamd_ffi_model_set_f32": [_vp, _i32, _i32, _i32],
    "beatamd_seis_gflib_device_ptr": [_vp, _i32, C.POINTER(_vp)],
    "beatamd_seis_gflib_destroy": [_vp, _i32],
    "beatamd_seis_stack_all_batch": [_vp, _i32, _i64, _vp, _vp, _vp, _i32, _vp],
    "beatamd_geo_gflib_create": [_vp, _i64, _i64, _vp, _pi32],
    "beatamd_geo_gflib_destroy": [_vp, _i32],
    "beatamd_geo_stack_all_batch": [_vp, _i32, _i64, _vp, _i32, _vp],
    "beatamd_weights_create": [_vp, _i32, _i64, _i64, _vp, _vp, _pi32],
    "beatamd_weights_update": [_vp, _i32, _i32, _i64, _vp, _vp],
    "beatamd_weights_destroy": [_vp, _i32],
    "beatamd_weights_band": [_vp, _i32, _pi64],
    "beatamd_weights_band_info": [_vp, _i32, _pi64, C.POINTER(_f64)],
    "beatamd_mvn_chol_logp_batch": [_vp, _i32, _i64, _vp, _vp, _vp],
    "beatamd_laplacian_create": [_vp, _i64, _vp, _f64, _pi32],
    "beatamd_laplacian_destroy": [_vp, _i32],
    "beatamd_laplacian_logp_batch": [_vp, _i32, _i64, _i64, _vp, _vp, _vp],
    "beatamd_ffi_model_create": [_vp, C.POINTER(FfiLayout), _i32, _vp, _vp, _vp, _pi32],
    "beatamd_ffi_model_add_wavemap": [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32],
    "beatamd_ffi_model_add_geodetic": [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "beatamd_ffi_model_add_geodetic_geometry": [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f64,
                                                _vp, _vp, _i32, _vp, _vp, _vp],
    "beatamd_ffi_model_set_laplacian": [_vp, _i32, _i32],
    "beatamd_ffi_model_nllk": [_vp, _i32, _pi64],
    "beatamd_ffi_model_destroy": [_vp, _i32],
    "beatamd_ffi_logp_batch": [_vp, _i32, _i64, _vp, _vp],
    "beatamd_ffi_synthetics_batch": [_vp, _i32, _i32, _i64, _vp, _i32, _vp],
    "beatamd_ffi_astep_batch": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp],
    "beatamd_autocovariance_batch": [_vp, _i64, _i64, _vp, _vp, _vp],
    "beatamd_scaled_toeplitz_batch": [_vp, _i64, _i64, _vp, _vp, _vp],
    "beatamd_ffi_astep_batch_betas": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "beatamd_ffi_mstep_batch": [_vp, _i32, _i64, _vp, _vp, _vp, _i64, _i32, _i32, C.c_uint64, C.c_uint32, _i64,
                                _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp],
    "beatamd_ctx_last_kernel": [_vp, C.c_char_p, _i64],
    "beatamd_ctx_gf_group_stats": [_vp, _pi64, C.POINTER(_f64), _pi64, _pi64],
    "beatamd_ctx_gf_plan": [_vp, C.c_char_p, _i64, C.POINTER(_f64), _pi64],
    "beatamd_ctx_gf_tune_log": [_vp, C.c_char_p, _i64],
    "beatamd_gf_patch_ranges": [_i64, _i64, _i64, C.c_int32],
    "beatamd_ctx_reload_knobs": [_vp],
    "beatamd_ctx_gf_chain_groups": [_vp, _i64, _vp, _vp, _i64, _vp],
    "beatamd_smc_calc_beta": [_vp, _i64, _vp, _i64, _f64, _f64, C.POINTER(_f64), _vp],
    "beatamd_smc_stage_weights": [_vp, _i64, _vp, _i64, _f64, _vp],
    "beatamd_smc_resample": [_vp, _i64, _vp, _f64, _vp],
    "beatamd_smc_population_factor": [_vp, _i64, _i64, _vp, _vp, _vp],
    "beatamd_proposal_draw": [_vp, _i64, _i64, _i64, _vp, C.c_uint64, C.c_uint32, _i64, _i32, _vp, _vp],
    "beatamd_proposal_draw_univariate": [_vp, _i64, _i64, _i32, _vp, C.c_uint64, C.c_uint32, _i64, _vp, _vp],
    "beatamd_gather_rows": [_vp, _i64, _i64, _vp, _i64, _vp, _vp],
    "beatamd_metropolis_tune": [_vp, _i64, _vp, _vp, _i32],
    "beatamd_like_assemble": [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp],
    "beatamd_metropolis_propose": [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "beatamd_metropolis_accept": [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp],
    "beatamd_whiten_rows": [_vp, _vp, _i64, _i64, _vp],
    "beatamd_whiten_rows_batch": [_vp, _vp, _i64, _i64, _i64, _vp],
    "beatamd_chol_inverse_batch": [_vp, _i64, _i64, _vp, _vp, _vp],
    "beatamd_chol_inverse_batch_flags": [_vp, _i64, _i64, _vp, _vp, _vp, _vp],
    "beatamd_whitening_ratio_batch": [_vp, _i64, _i64, _vp, _vp, _vp],
    "beatamd_unwhiten_traces": [_vp, _i64, _i64, _vp, _vp],
    "beatamd_factor_compact": [_vp, _i64, _i64, _vp, _vp],
    "beatamd_ffi_model_update_data": [_vp, _i32, _i32, _vp],
    "beatamd_halfspace_displacements_batch": [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _f64, _vp],
}

EXPORTS = sorted(list(_PROTOS) + ["beatamd_last_error", "beatamd_version"])

_lib = None


def load():
    """Load libbeat_amd.so (needs only the HIP runtime, not a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BeatAmdError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C beat_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7; if
        # torch is imported AFTER a library bound to /opt/rocm's copy, torch finds "No HIP
        # GPUs".  Importing torch first lets the loader resolve our DT_NEEDED libamdhip64.so.7
        # to the copy torch already mapped.  Without torch the system runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, args in _PROTOS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.beatamd_last_error.restype = C.c_char_p
        lib.beatamd_last_error.argtypes = []
        lib.beatamd_version.restype = C.c_int
        if lib.beatamd_version() != ABI_VERSION:
            raise BeatAmdError("libbeat_amd.so has ABI revision %d, beat_amd expects %d: rebuild with "
                               "`make -C beat_amd/csrc`" % (lib.beatamd_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc):
    """Map a status code to the exception the reference raises in that situation."""
    if rc == OK:
        return
    msg = load().beatamd_last_error().decode("utf-8", "replace")
    if rc in (EINVAL, EBADCOV):
        raise ValueError(msg)
    if rc == EINDEX:
        raise IndexError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    if rc == ENOTPSD:
        raise np.linalg.LinAlgError(msg)
    raise BeatAmdError(msg)


def ptr(a):
    """void* of a numpy array (host), a torch tensor (host or device) or a raw int."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError("cannot take a pointer of %r" % type(a))


def f64(a):
    """C-contiguous float64 view/copy of a host array; device tensors pass through."""
    if hasattr(a, "data_ptr") and not isinstance(a, np.ndarray):
        import torch
        if a.dtype != torch.float64 or not a.is_contiguous():
            raise ValueError("device tensors must be contiguous float64")
        return a
    return np.ascontiguousarray(a, dtype=np.float64)
