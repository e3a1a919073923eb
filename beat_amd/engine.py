"""
Thin object layer over the C ABI: one :class:`Context` per (process, GPU).

Arrays may be numpy arrays (host; results come back as numpy) or torch CUDA tensors
(device; results are torch tensors on the same device, the call is asynchronous on the
context stream).  PyTorch is only plumbing here (device memory, streams); all arithmetic
is in ``libbeat_amd.so``.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import INTERPOLATIONS, check, f64, ptr


def _is_dev(a):
    return hasattr(a, "data_ptr") and not isinstance(a, np.ndarray) and a.is_cuda


def _empty_like(ref, shape, dtype=np.float64):
    if _is_dev(ref):
        import torch
        tdt = {np.float64: torch.float64, np.int32: torch.int32}[dtype]
        return torch.empty(shape, dtype=tdt, device=ref.device)
    return np.empty(shape, dtype=dtype)


def _i32(a, ref=None):
    if _is_dev(a):
        import torch
        if a.dtype != torch.int32 or not a.is_contiguous():
            raise ValueError("device index tensors must be contiguous int32")
        return a
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64arr(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def _interp(interpolation):
    if interpolation not in INTERPOLATIONS:
        raise NotImplementedError("Interpolation scheme %s not implemented!" % interpolation)
    return INTERPOLATIONS[interpolation]


class Context(object):
    """Owns the HBM-resident state of one GPU: GF libraries, weights, models, scratch."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.beatamd_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.beatamd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _adopt_stream(self, *arrays):
        """Device tensors are produced on torch's current stream: launch there too, so the
        kernels are ordered after their producers without an explicit synchronisation."""
        for a in arrays:
            if _is_dev(a):
                import torch
                s = torch.cuda.current_stream(self.device).cuda_stream
                if s != getattr(self, "_stream", None):
                    check(self._lib.beatamd_ctx_set_stream(self._h, C.c_void_p(s)))
                    self._stream = s
                return

    def use_torch_stream(self):
        """Launch on torch's current stream of this device (so torch ops interleave in order)."""
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self._lib.beatamd_ctx_set_stream(self._h, C.c_void_p(s)))
        self._stream = s

    def set_stream(self, stream_ptr):
        """stream_ptr 0 / None = the HIP null stream (torch's default stream)"""
        check(self._lib.beatamd_ctx_set_stream(self._h, C.c_void_p(stream_ptr or 0)))
        self._stream = stream_ptr or 0

    def use_own_stream(self):
        check(self._lib.beatamd_ctx_use_own_stream(self._h))
        self._stream = None

    def synchronize(self):
        check(self._lib.beatamd_ctx_synchronize(self._h))

    def set_step_counter(self, counter):
        """counter: torch int32 tensor (1,) on the device, or None -- see beatamd_ctx_set_step_counter"""
        check(self._lib.beatamd_ctx_set_step_counter(self._h, ptr(counter) if counter is not None else None))

    def enable_timing(self, on=True):
        check(self._lib.beatamd_ctx_enable_timing(self._h, int(bool(on))))

    def reset_timing(self):
        check(self._lib.beatamd_ctx_reset_timing(self._h))

    def kernel_time(self, name):
        """-> (total_ms, launches) measured with HIP events on the launch stream"""
        ms, n = C.c_double(), C.c_int64()
        check(self._lib.beatamd_ctx_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- fast sweep
    def fast_sweep_batch(self, slowness, patch_size, h_strk, h_dip, num_strk, num_dip):
        self._adopt_stream(slowness)
        slowness = f64(slowness)
        n = int(num_strk) * int(num_dip)
        Cn = int(slowness.shape[0]) if slowness.ndim == 2 else 1
        if slowness.ndim == 1:
            slowness = slowness.reshape(1, -1)
        if slowness.shape[1] != n:
            raise ValueError("slowness has %d entries per chain, grid has %d" % (slowness.shape[1], n))
        hs, hd = _i32(h_strk), _i32(h_dip)
        out = _empty_like(slowness, (Cn, n))
        check(self._lib.beatamd_fast_sweep_batch(self._h, ptr(slowness), float(patch_size), ptr(hs),
                                                 ptr(hd), int(num_strk), int(num_dip), Cn, ptr(out)))
        return out

    # -- seismic GF library
    def seis_gflib_create(self, dims, starttime_min, starttime_sampling, duration_min,
                          duration_sampling):
        T, P, D, S, N = [int(d) for d in dims]
        lid = C.c_int32()
        check(self._lib.beatamd_seis_gflib_create(self._h, T, P, D, S, N, float(starttime_min),
                                                  float(starttime_sampling), float(duration_min),
                                                  float(duration_sampling), C.byref(lid)))
        return lid.value

    def seis_gflib_upload(self, lib_id, array, offset=0):
        a = f64(array)
        count = int(a.numel()) if _is_dev(a) else int(a.size)
        check(self._lib.beatamd_seis_gflib_upload(self._h, lib_id, ptr(a), int(offset), count))

    def seis_gflib_adopt(self, lib_id, device_tensor):
        self._adopt_stream(device_tensor)
        check(self._lib.beatamd_seis_gflib_adopt(self._h, lib_id, ptr(device_tensor)))

    def seis_gflib_round_to_f32(self, lib_id):
        """float copy of the library + the float64 storage rounded to the same values"""
        check(self._lib.beatamd_seis_gflib_round_to_f32(self._h, lib_id))

    def ffi_model_set_f32(self, model_id, wavemap_index, on=True):
        check(self._lib.beatamd_ffi_model_set_f32(self._h, model_id, int(wavemap_index), 1 if on else 0))

    def seis_gflib_device_ptr(self, lib_id):
        p = C.c_void_p()
        check(self._lib.beatamd_seis_gflib_device_ptr(self._h, lib_id, C.byref(p)))
        return p.value

    def seis_gflib_destroy(self, lib_id):
        check(self._lib.beatamd_seis_gflib_destroy(self._h, lib_id))

    def seis_stack_all_batch(self, lib_id, dims, durations, starttimes, slips,
                             interpolation="nearest_neighbor"):
        self._adopt_stream(durations, starttimes, slips)
        T, P, D, S, N = dims
        it = _interp(interpolation)
        du, st, sl = f64(durations), f64(starttimes), f64(slips)
        Cn = int(du.shape[0])
        if tuple(du.shape) != (Cn, P) or tuple(sl.shape) != (Cn, P) or tuple(st.shape) != (Cn, T, P):
            raise ValueError("stack_all_batch: expected durations/slips (C,%d) and starttimes (C,%d,%d)"
                             % (P, T, P))
        out = _empty_like(du, (Cn, T, N))
        check(self._lib.beatamd_seis_stack_all_batch(self._h, lib_id, Cn, ptr(du), ptr(st), ptr(sl), it,
                                                     ptr(out)))
        return out

    # -- geodetic GF library
    def geo_gflib_create(self, G):
        G = f64(G)
        lid = C.c_int32()
        check(self._lib.beatamd_geo_gflib_create(self._h, int(G.shape[0]), int(G.shape[1]), ptr(G),
                                                 C.byref(lid)))
        return lid.value

    def geo_gflib_destroy(self, lib_id):
        check(self._lib.beatamd_geo_gflib_destroy(self._h, lib_id))

    def geo_stack_all_batch(self, lib_id, nobs, slips, out=None):
        self._adopt_stream(slips)
        sl = f64(slips)
        Cn = int(sl.shape[0])
        acc = out is not None
        if out is None:
            out = _empty_like(sl, (Cn, nobs))
        check(self._lib.beatamd_geo_stack_all_batch(self._h, lib_id, Cn, ptr(sl), int(acc), ptr(out)))
        return out

    # -- weights / likelihood
    def weights_create_scalar(self, w, slog_pdet, M):
        w, sl = f64(w).ravel(), f64(slog_pdet).ravel()
        wid = C.c_int32()
        check(self._lib.beatamd_weights_create(self._h, _lib.W_SCALAR, int(w.size), int(M), ptr(w),
                                               ptr(sl), C.byref(wid)))
        return wid.value

    def weights_create_dense(self, W, slog_pdet):
        W, sl = f64(W), f64(slog_pdet).ravel()
        if W.ndim == 2:
            W = W.reshape((1,) + W.shape)
        nd, M, M2 = W.shape
        if M != M2:
            raise ValueError("weights must be square")
        wid = C.c_int32()
        check(self._lib.beatamd_weights_create(self._h, _lib.W_DENSE, int(nd), int(M), ptr(W), ptr(sl),
                                               C.byref(wid)))
        return wid.value

    def weights_update(self, wset_id, weights, slog_pdet):
        """new weights of an existing set; scalar (nd,) or dense (nd, M, M) -- kind and size are
        checked against the set by the library"""
        W, sl = f64(weights), f64(slog_pdet).ravel()
        kind = _lib.W_DENSE if W.ndim >= 2 else _lib.W_SCALAR
        count = int(W.numel()) if _is_dev(W) else int(W.size)
        check(self._lib.beatamd_weights_update(self._h, wset_id, kind, count, ptr(W), ptr(sl)))

    def weights_band(self, wset_id):
        """half bandwidth a dense weight set is evaluated on (banded upper-triangular whitening operators, e.g. the
        bidiagonal ones of the reference's "exponential" noise structure), -1: the dense kernel"""
        b = C.c_int64()
        check(self._lib.beatamd_weights_band(self._h, wset_id, C.byref(b)))
        return b.value

    def weights_band_info(self, wset_id):
        """-> (half bandwidth or -1, largest entry beyond the band relative to the largest entry of its row): what the banded
        evaluation of this weight set leaves out (<= 2^-40 by construction)"""
        b, d = C.c_int64(), C.c_double()
        check(self._lib.beatamd_weights_band_info(self._h, wset_id, C.byref(b), C.byref(d)))
        return b.value, d.value

    def weights_destroy(self, wset_id):
        check(self._lib.beatamd_weights_destroy(self._h, wset_id))

    def mvn_chol_logp_batch(self, wset_id, residuals, hp):
        self._adopt_stream(residuals, hp)
        r, h = f64(residuals), f64(hp)
        Cn, nd = int(r.shape[0]), int(r.shape[1])
        out = _empty_like(r, (Cn, nd))
        check(self._lib.beatamd_mvn_chol_logp_batch(self._h, wset_id, Cn, ptr(r), ptr(h), ptr(out)))
        return out

    def laplacian_create(self, L, logdet):
        L = f64(L)
        lid = C.c_int32()
        check(self._lib.beatamd_laplacian_create(self._h, int(L.shape[0]), ptr(L), float(logdet),
                                                 C.byref(lid)))
        return lid.value

    def laplacian_destroy(self, lap_id):
        check(self._lib.beatamd_laplacian_destroy(self._h, lap_id))

    def laplacian_logp_batch(self, lap_id, slips, hp):
        self._adopt_stream(slips, hp)
        s, h = f64(slips), f64(hp)
        Cn, nvar = int(s.shape[0]), int(s.shape[1])
        out = _empty_like(s, (Cn,))
        check(self._lib.beatamd_laplacian_logp_batch(self._h, lap_id, Cn, nvar, ptr(s), ptr(h), ptr(out)))
        return out

    # -- noise covariance estimation
    def autocovariance_batch(self, data):
        d = f64(data)
        self._adopt_stream(d)
        if _is_dev(d):
            mean = d.mean(dim=1).contiguous()
        else:
            mean = np.ascontiguousarray(d.mean(axis=1))
        out = _empty_like(d, tuple(d.shape))
        check(self._lib.beatamd_autocovariance_batch(self._h, int(d.shape[0]), int(d.shape[1]), ptr(d),
                                                     ptr(mean), ptr(out)))
        return out

    def scaled_toeplitz_batch(self, coeffs, stds):
        c, s = f64(coeffs), f64(stds)
        self._adopt_stream(c, s)
        nd, n = int(c.shape[0]), int(c.shape[1])
        out = _empty_like(c, (nd, n, n))
        check(self._lib.beatamd_scaled_toeplitz_batch(self._h, nd, n, ptr(c), ptr(s), ptr(out)))
        return out

    # -- fused FFI model
    def ffi_model_create(self, layout, n_patch_dip, n_patch_strike, patch_size):
        nd = np.ascontiguousarray(n_patch_dip, dtype=np.int32)
        ns = np.ascontiguousarray(n_patch_strike, dtype=np.int32)
        ps = np.ascontiguousarray(patch_size, dtype=np.float64)
        mid = C.c_int32()
        check(self._lib.beatamd_ffi_model_create(self._h, C.byref(layout), int(nd.size), ptr(nd) if nd.size else None,
                                                 ptr(ns) if ns.size else None, ptr(ps) if ps.size else None,
                                                 C.byref(mid)))
        return mid.value

    def ffi_model_add_wavemap(self, model_id, lib_ids, data, wset_id, hp_off, shift_off=None,
                              interpolation="nearest_neighbor"):
        libs = np.ascontiguousarray(lib_ids, dtype=np.int32)
        data = np.ascontiguousarray(data, dtype=np.float64)
        hp = _i64arr(hp_off)
        sh = _i64arr(shift_off)
        check(self._lib.beatamd_ffi_model_add_wavemap(self._h, model_id, ptr(libs), ptr(data), wset_id,
                                                      ptr(hp), ptr(sh), _interp(interpolation)))

    def ffi_model_add_geodetic(self, model_id, geo_lib_ids, data, odws, sizes, wset_ids, hp_off):
        libs = np.ascontiguousarray(geo_lib_ids, dtype=np.int32)
        data = np.ascontiguousarray(data, dtype=np.float64)
        odws = np.ascontiguousarray(odws, dtype=np.float64)
        sizes = _i64arr(sizes)
        ws = np.ascontiguousarray(wset_ids, dtype=np.int32)
        hp = _i64arr(hp_off)
        check(self._lib.beatamd_ffi_model_add_geodetic(self._h, model_id, ptr(libs), ptr(data), ptr(odws),
                                                       int(sizes.size), ptr(sizes), ptr(ws), ptr(hp)))

    def ffi_model_add_geodetic_geometry(self, model_id, kinds, param_off, param_fixed, east, north,
                                        los, nu, data, odws, sizes, wset_ids, hp_off):
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        po = np.ascontiguousarray(param_off, dtype=np.int64)
        pf = np.ascontiguousarray(param_fixed, dtype=np.float64)
        east = np.ascontiguousarray(east, dtype=np.float64)
        north = np.ascontiguousarray(north, dtype=np.float64)
        los = np.ascontiguousarray(los, dtype=np.float64)
        data = np.ascontiguousarray(data, dtype=np.float64)
        odws = np.ascontiguousarray(odws, dtype=np.float64)
        sizes = _i64arr(sizes)
        ws = np.ascontiguousarray(wset_ids, dtype=np.int32)
        hp = _i64arr(hp_off)
        check(self._lib.beatamd_ffi_model_add_geodetic_geometry(
            self._h, model_id, int(kinds.size), ptr(kinds), ptr(po), ptr(pf), int(east.size), ptr(east),
            ptr(north), ptr(los), float(nu), ptr(data), ptr(odws), int(sizes.size), ptr(sizes), ptr(ws),
            ptr(hp)))

    def ffi_model_set_laplacian(self, model_id, lap_id):
        check(self._lib.beatamd_ffi_model_set_laplacian(self._h, model_id, lap_id))

    def ffi_model_nllk(self, model_id):
        n = C.c_int64()
        check(self._lib.beatamd_ffi_model_nllk(self._h, model_id, C.byref(n)))
        return n.value

    def ffi_model_destroy(self, model_id):
        check(self._lib.beatamd_ffi_model_destroy(self._h, model_id))

    def ffi_logp_batch(self, model_id, Q, nllk, out=None):
        self._adopt_stream(Q)
        Q = f64(Q)
        Cn = int(Q.shape[0])
        if out is None:
            out = _empty_like(Q, (Cn, nllk))
        check(self._lib.beatamd_ffi_logp_batch(self._h, model_id, Cn, ptr(Q), ptr(out)))
        return out

    def ffi_synthetics_batch(self, model_id, wavemap_index, Q, T, N, residuals=False, out=None):
        """synthetics (or data - synthetics) [C, T, N] of one wavemap at the points Q [C, nparams]"""
        self._adopt_stream(Q)
        Qc = f64(Q)
        Cn = int(Qc.shape[0])
        if out is None:
            out = _empty_like(Qc, (Cn, int(T), int(N)))
        check(self._lib.beatamd_ffi_synthetics_batch(self._h, model_id, int(wavemap_index), Cn, ptr(Qc),
                                                     int(bool(residuals)), ptr(out)))
        return out

    def ffi_astep_batch(self, model_id, Q0, L0, delta, scaling, lower, upper, log_u, beta,
                        accepted=None):
        """In-place update of Q0 / L0 (must be contiguous float64); returns accepted (int32)."""
        self._adopt_stream(Q0, delta)
        Cn = int(Q0.shape[0])
        if accepted is None:
            accepted = _empty_like(Q0, (Cn,), np.int32)
        for a in (Q0, L0):
            if isinstance(a, np.ndarray) and not (a.flags.c_contiguous and a.dtype == np.float64):
                raise ValueError("Q0 / L0 must be C-contiguous float64 (updated in place)")
        # converted arrays are bound to locals: a temporary would be freed before the call reads it
        de, sc, lo, up, lu = f64(delta), f64(scaling), f64(lower), f64(upper), f64(log_u)
        if np.ndim(beta) == 0 and not hasattr(beta, "data_ptr"):
            check(self._lib.beatamd_ffi_astep_batch(self._h, model_id, Cn, ptr(Q0), ptr(L0), ptr(de),
                                                    ptr(sc), ptr(lo), ptr(up), ptr(lu), float(beta),
                                                    ptr(accepted)))
        else:  # one beta per chain (parallel tempering replicas)
            be = f64(beta)
            check(self._lib.beatamd_ffi_astep_batch_betas(self._h, model_id, Cn, ptr(Q0), ptr(L0),
                                                          ptr(de), ptr(sc), ptr(lo), ptr(up), ptr(lu),
                                                          ptr(be), ptr(accepted)))
        return accepted

    def ffi_mstep_batch(self, model_id, Q0, L0, factor, kind, df, seed, step, first_chain, scaling, lower,
                        upper, beta, accepted, accepted_sum=None, n_accepted=None):
        """One Metropolis step of every chain with the proposal drawn on the device
        (beatamd_ffi_mstep_batch): device tensors only, nothing returns to the host.
        kind None / -1: multivariate proposal with ``factor`` [K, nparams] (df > 0: multivariate t);
        0/1/2: per-parameter Normal / Cauchy / Laplace with ``factor`` = scales [nparams]."""
        self._adopt_stream(Q0, L0)
        Cn = int(Q0.shape[0])
        kind = -1 if kind is None else int(kind)
        fa, sc, lo, up = f64(factor), f64(scaling), f64(lower), f64(upper)
        K = int(fa.shape[0]) if kind < 0 else 0
        scalar = np.ndim(beta) == 0 and not hasattr(beta, "data_ptr")
        be = None if scalar else f64(beta)
        check(self._lib.beatamd_ffi_mstep_batch(
            self._h, model_id, Cn, ptr(Q0), ptr(L0), ptr(fa), K, kind, int(df), int(seed) & (2 ** 64 - 1),
            int(step) & 0xffffffff, int(first_chain), ptr(sc), ptr(lo), ptr(up), float(beta) if scalar else 1.0,
            None if scalar else ptr(be), ptr(accepted), None if accepted_sum is None else ptr(accepted_sum),
            None if n_accepted is None else ptr(n_accepted)))
        return accepted

    # -- introspection
    def last_kernel(self):
        """name<template arguments> of the stacking kernel the most recent call launched"""
        buf = C.create_string_buffer(128)
        check(self._lib.beatamd_ctx_last_kernel(self._h, buf, 128))
        return buf.value.decode()

    def gf_group_stats(self):
        """-> dict(chains_per_group, mean_rows, max_rows, row_bytes) of the last chain-shared launch"""
        cg, mx, rb, mean = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        check(self._lib.beatamd_ctx_gf_group_stats(self._h, C.byref(cg), C.byref(mean), C.byref(mx),
                                                   C.byref(rb)))
        return dict(chains_per_group=cg.value, mean_rows=mean.value, max_rows=mx.value,
                    row_bytes=rb.value)

    def reload_knobs(self):
        """read the BEATAMD_G* / BEATAMD_WS_MAP / BEATAMD_SWEEP_V1 knobs from the environment again (they are read once,
        when the context is created, unless it was created under BEATAMD_KNOBS_LIVE=1)"""
        check(self._lib.beatamd_ctx_reload_knobs(self._h))

    def gf_plan(self, passes=True):
        """-> dict(plan=<what the kernel selection chose for the last stacking launch and why>, mean_passes, max_passes:
        row passes per (chain group, target, patch); 1 unless a patch touched more rows than an LDS buffer holds)"""
        buf = C.create_string_buffer(512)
        mean, mx = C.c_double(), C.c_int64()
        check(self._lib.beatamd_ctx_gf_plan(self._h, buf, 512, C.byref(mean) if passes else None,
                                            C.byref(mx) if passes else None))
        out = dict(plan=buf.value.decode())
        if passes:
            out.update(mean_passes=mean.value, max_passes=mx.value)
        return out

    def gf_tune_log(self):
        """the most recent group-size measurement of the lane <-> chain stacking kernels, in words (empty before the
        first one): the first call of a problem shape launches every candidate group size twice and keeps the fastest"""
        buf = C.create_string_buffer(512)
        check(self._lib.beatamd_ctx_gf_tune_log(self._h, buf, 512))
        return buf.value.decode()

    def gf_chain_groups(self, key0, key1, chains_per_group=512):
        """how a batch is cut into its chain groups (scheduling only): key0 / key1 (C,) per-chain keys -- the hypocentre
        (strike, dip) in the fused model path -> uint32 array [ngroups, chains_per_group] of chain ids, 0xffffffff behind
        the last chain (recursive bisection along the key of the wider extent, k_gc_cut)"""
        import torch
        dev = "cuda:%d" % self.device
        k0, k1 = [k if _is_dev(k) else torch.as_tensor(np.ascontiguousarray(k, dtype=np.float64)).to(dev) for k in (key0, key1)]
        self._adopt_stream(k0, k1)
        k0, k1 = f64(k0), f64(k1)
        Cn = int(k0.numel())
        ng = (Cn + chains_per_group - 1) // chains_per_group
        out = np.empty(ng * chains_per_group, dtype=np.uint32)
        check(self._lib.beatamd_ctx_gf_chain_groups(self._h, Cn, ptr(k0), ptr(k1), int(chains_per_group),
                                                    out.ctypes.data_as(C.c_void_p)))
        return out.reshape(ng, chains_per_group)

    # -- sampler steps on the device (SMC stage transition, proposals, exchange)
    def smc_calc_beta(self, likelihoods, beta, coef_variation, stride=1, n=None):
        """smc.py:133-165 -> (beta_new, weights); likelihoods: (C,) array or a strided view
        described by (tensor, stride, n)"""
        self._adopt_stream(likelihoods)
        lk = f64(likelihoods)
        Cn = int(n if n is not None else (lk.numel() if _is_dev(lk) else lk.size))
        w = _empty_like(lk, (Cn,))
        b = C.c_double()
        check(self._lib.beatamd_smc_calc_beta(self._h, Cn, ptr(lk), int(stride), float(beta),
                                              float(coef_variation), C.byref(b), ptr(w)))
        return b.value, w

    def smc_stage_weights(self, likelihoods, dbeta, stride=1, n=None):
        self._adopt_stream(likelihoods)
        lk = f64(likelihoods)
        Cn = int(n if n is not None else (lk.numel() if _is_dev(lk) else lk.size))
        w = _empty_like(lk, (Cn,))
        check(self._lib.beatamd_smc_stage_weights(self._h, Cn, ptr(lk), int(stride), float(dbeta), ptr(w)))
        return w

    def smc_resample(self, weights, aux):
        """smc.py:290-324 -> resampling indexes (int32)"""
        self._adopt_stream(weights)
        w = f64(weights)
        Cn = int(w.numel()) if _is_dev(w) else int(w.size)
        idx = _empty_like(w, (Cn,), np.int32)
        check(self._lib.beatamd_smc_resample(self._h, Cn, ptr(w), float(aux), ptr(idx)))
        return idx

    def smc_population_factor(self, population, weights):
        self._adopt_stream(population, weights)
        X, w = f64(population), f64(weights)
        Cn, npar = int(X.shape[0]), int(X.shape[1])
        F = _empty_like(X, (Cn, npar))
        check(self._lib.beatamd_smc_population_factor(self._h, Cn, npar, ptr(X), ptr(w), ptr(F)))
        return F

    def proposal_draw(self, factor, n_chains, seed, step, first_chain=0, df=0, delta=None, log_u=None,
                      want_log_u=True):
        """delta (n_chains, nparams) = z . factor with z from Philox4x32-10; -> (delta, log_u)"""
        self._adopt_stream(factor)
        F = f64(factor)
        K, npar = int(F.shape[0]), int(F.shape[1])
        if delta is None:
            delta = _empty_like(F, (int(n_chains), npar))
        if log_u is None and want_log_u:
            log_u = _empty_like(F, (int(n_chains),))
        check(self._lib.beatamd_proposal_draw(self._h, int(n_chains), K, npar, ptr(F), int(seed) & (2 ** 64 - 1),
                                              int(step) & 0xffffffff, int(first_chain), int(df), ptr(delta),
                                              ptr(log_u)))
        return delta, log_u

    def proposal_draw_univariate(self, kind, scale, n_chains, seed, step, first_chain=0, delta=None, log_u=None,
                                 want_log_u=True):
        """delta (n_chains, nparams): independent Normal (0) / Cauchy (1) / Laplace (2) draws per
        component times scale[j] (beat/sampler/base.py:129-160); -> (delta, log_u)"""
        self._adopt_stream(scale)
        sc = f64(scale)
        npar = int(sc.shape[0])
        if delta is None:
            delta = _empty_like(sc, (int(n_chains), npar))
        if log_u is None and want_log_u:
            log_u = _empty_like(sc, (int(n_chains),))
        check(self._lib.beatamd_proposal_draw_univariate(self._h, int(n_chains), npar, int(kind), ptr(sc),
                                                         int(seed) & (2 ** 64 - 1), int(step) & 0xffffffff,
                                                         int(first_chain), ptr(delta), ptr(log_u)))
        return delta, log_u

    def gather_rows(self, src, indexes, out=None):
        self._adopt_stream(src, indexes)
        S = f64(src)
        idx = _i32(indexes)
        nout = int(idx.numel()) if _is_dev(idx) else int(idx.size)
        if out is None:
            out = _empty_like(S, (nout, int(S.shape[1])))
        check(self._lib.beatamd_gather_rows(self._h, nout, int(S.shape[1]), ptr(S), int(S.shape[0]),
                                            ptr(idx), ptr(out)))
        return out

    def metropolis_tune(self, scaling, accepted, tune_interval):
        """in place: scaling *= pymc tune factor(accepted / interval); accepted = 0"""
        self._adopt_stream(scaling)
        check(self._lib.beatamd_metropolis_tune(self._h, int(scaling.shape[0]), ptr(scaling), ptr(accepted),
                                                int(tune_interval)))

    # -- a Metropolis step in pieces (a collective between forward model and acceptance: target-sharded models)
    def like_assemble(self, gathered, dst_col, local_ll, local_col0, n_rest, rest_dst0, group_end, LL):
        """LL [C, nllk] (device) from the all-gathered rows `gathered` [nsrc, C] of all ranks: row r -> column dst_col[r]
        (-1: a rank's flag row, NaN = chain outside the library grid there), the replicated columns
        local_ll[:, local_col0 : local_col0 + n_rest] -> columns rest_dst0.., like = the composites' sums (k_like_sum's order)"""
        self._adopt_stream(gathered, LL)
        dc = np.ascontiguousarray(dst_col, dtype=np.int32)
        ge = np.ascontiguousarray(group_end, dtype=np.int32)
        Cn, nllk = int(LL.shape[0]), int(LL.shape[1])
        if int(gathered.shape[0]) != dc.size or int(gathered.shape[1]) != Cn or not gathered.is_contiguous() or not LL.is_contiguous():
            raise ValueError("like_assemble: gathered must be a contiguous (nsrc, C) tensor, LL a contiguous (C, nllk) one")
        check(self._lib.beatamd_like_assemble(self._h, Cn, nllk, dc.size, ptr(gathered), dc.ctypes.data,
                                              ptr(local_ll) if n_rest else None, int(local_ll.stride(0)) if n_rest else 0,
                                              int(local_col0), int(n_rest), int(rest_dst0), ge.size, ge.ctypes.data, ptr(LL)))
        return LL

    def metropolis_propose(self, Q0, delta, scaling, lower, upper, Qprop, inbounds):
        """metropolis.py:313-343 for all chains: Qprop = Q0 + delta * scaling inside the prior box, else Q0; inbounds int32 [C]"""
        self._adopt_stream(Q0, Qprop)
        check(self._lib.beatamd_metropolis_propose(self._h, int(Q0.shape[0]), int(Q0.shape[1]), ptr(f64(Q0)), ptr(f64(delta)),
                                                   ptr(f64(scaling)), ptr(f64(lower)), ptr(f64(upper)), ptr(Qprop), ptr(inbounds)))

    def metropolis_accept(self, Q0, L0, Qprop, Lprop, inbounds, log_u, beta, accepted):
        """metropolis.py:344-385 for all chains, in place on Q0 / L0; beta: float or a device tensor [C]"""
        self._adopt_stream(Q0, L0)
        per_chain = hasattr(beta, "data_ptr")
        check(self._lib.beatamd_metropolis_accept(self._h, int(Q0.shape[0]), int(Q0.shape[1]), int(L0.shape[1]), ptr(Q0), ptr(L0),
                                                  ptr(Qprop), ptr(Lprop), ptr(inbounds), ptr(f64(log_u)),
                                                  1.0 if per_chain else float(beta), ptr(f64(beta)) if per_chain else None,
                                                  ptr(accepted)))

    def halfspace_displacements_batch(self, kinds, params, east, north, nu=0.25):
        """params (C, nsrc, 10) -> (C, nsrc, nobs, 3) = (north, east, up) [m]"""
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        prm = f64(params)
        self._adopt_stream(prm)
        Cn, nsrc = int(prm.shape[0]), int(prm.shape[1])
        if tuple(prm.shape[1:]) != (kinds.size, 10):
            raise ValueError("params must be (C, %d, 10)" % kinds.size)
        e, n = f64(east), f64(north)
        nobs = int(e.numel()) if _is_dev(e) else int(e.size)
        out = _empty_like(prm, (Cn, nsrc, nobs, 3))
        check(self._lib.beatamd_halfspace_displacements_batch(self._h, Cn, nsrc, ptr(kinds), ptr(prm), nobs,
                                                              ptr(e), ptr(n), float(nu), ptr(out)))
        return out

    def chol_inverse_batch(self, covs):
        """covs (nd, n, n) -> (W (nd, n, n) upper triangular = cholesky(inv(C)).T, log_pdet (nd,));
        numpy or torch-cuda in, same kind out; numpy.linalg.LinAlgError if not positive definite"""
        if _is_dev(covs):
            self._adopt_stream(covs)
        C = f64(covs)
        nd, n = int(C.shape[0]), int(C.shape[1])
        W = _empty_like(C, (nd, n, n))
        lp = _empty_like(C, (nd,))
        check(self._lib.beatamd_chol_inverse_batch(self._h, nd, n, ptr(C), ptr(W), ptr(lp)))
        return W, lp

    def chol_inverse_batch_flags(self, covs):
        """like chol_inverse_batch, but a matrix that is not positive definite is reported in the
        returned int32 flags (nd,) instead of raising: -> (W, log_pdet, not_psd)"""
        if _is_dev(covs):
            self._adopt_stream(covs)
        C = f64(covs)
        nd, n = int(C.shape[0]), int(C.shape[1])
        W = _empty_like(C, (nd, n, n))
        lp = _empty_like(C, (nd,))
        if _is_dev(C):
            import torch
            bad = torch.empty((nd,), dtype=torch.int32, device=C.device)
        else:
            bad = np.empty((nd,), dtype=np.int32)
        check(self._lib.beatamd_chol_inverse_batch_flags(self._h, nd, n, ptr(C), ptr(W), ptr(lp), ptr(bad)))
        return W, lp, bad

    def factor_compact(self, factor):
        """tall proposal factor (K, n) -> upper-triangular R (n, n) with R^T R = factor^T factor, or
        None when the Gram matrix is not numerically positive definite"""
        if _is_dev(factor):
            self._adopt_stream(factor)
        F = f64(factor)
        K, n = int(F.shape[0]), int(F.shape[1])
        R = _empty_like(F, (n, n))
        try:
            check(self._lib.beatamd_factor_compact(self._h, K, n, ptr(F), ptr(R)))
        except np.linalg.LinAlgError:
            return None
        return R

    def whitening_ratio_batch(self, W_new, W_old):
        """M (nd, n, n) = W_new . inv(W_old) for upper-triangular whitening operators"""
        Wn, Wo = f64(W_new), f64(W_old)
        if Wn.shape != Wo.shape or Wn.ndim != 3:
            raise ValueError("W_new and W_old must both be (nd, n, n)")
        M = _empty_like(Wn, tuple(Wn.shape))
        check(self._lib.beatamd_whitening_ratio_batch(self._h, int(Wn.shape[0]), int(Wn.shape[1]), ptr(Wn),
                                                      ptr(Wo), ptr(M)))
        return M

    def unwhiten_traces(self, W, X):
        """X (nd, n) <- inv(W[t]) . X[t] for upper-triangular W (nd, n, n), in place (back substitution)"""
        Wd = f64(W)
        if Wd.ndim != 3 or Wd.shape[1] != Wd.shape[2] or tuple(X.shape) != (Wd.shape[0], Wd.shape[1]):
            raise ValueError("W must be (nd, n, n) and X (nd, n)")
        if f64(X) is not X:
            raise ValueError("X must be a contiguous float64 array / tensor (updated in place)")
        check(self._lib.beatamd_unwhiten_traces(self._h, int(Wd.shape[0]), int(Wd.shape[1]), ptr(Wd), ptr(X)))
        return X

    def ffi_model_update_data(self, model_id, wavemap_index, data):
        d = f64(data)
        check(self._lib.beatamd_ffi_model_update_data(self._h, int(model_id), int(wavemap_index), ptr(d)))

    def whiten_rows_batch(self, rows, W):
        """rows (B, R, N) device tensor, in place: rows[b] <- rows[b] . W[b]^T for all datasets in one call
        (W (B, N, N) numpy or device; upper-triangular operators need no second buffer)"""
        self._adopt_stream(rows)
        if rows.dim() != 3 or not rows.is_contiguous():
            raise ValueError("whiten_rows_batch: rows must be a contiguous (B, R, N) device tensor")
        Wc = f64(W)
        if tuple(Wc.shape) != (rows.shape[0], rows.shape[2], rows.shape[2]):
            raise ValueError("whiten_rows_batch: W must be (%d, %d, %d)" % (rows.shape[0], rows.shape[2], rows.shape[2]))
        check(self._lib.beatamd_whiten_rows_batch(self._h, ptr(rows), int(rows.shape[0]), int(rows.shape[1]),
                                                  int(rows.shape[2]), ptr(Wc)))
        return rows

    def whiten_rows(self, rows, W):
        """rows (R, N) device tensor, in place: rows <- rows . W^T"""
        self._adopt_stream(rows)
        Wc = f64(W)
        check(self._lib.beatamd_whiten_rows(self._h, ptr(rows), int(rows.shape[0]), int(rows.shape[1]),
                                            ptr(Wc)))
        return rows


_contexts = {}


def get_context(device=None):
    """Process-wide context of a device (default: LOCAL_RANK or 0)."""
    import os
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    ctx = _contexts.get(device)
    if ctx is None or ctx._h is None:
        ctx = Context(device)
        _contexts[device] = ctx
    return ctx
