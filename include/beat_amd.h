/*
 * beat_amd.h -- C ABI of libbeat_amd.so: the MI355X (gfx950) forward-model +
 * likelihood engine for BEAT's SMC/PT inner loop.
 *
 * This is the drop-in boundary (DESIGN.md "Boundary"): plain pointers and sizes, no
 * torch / numpy / Python types.  Each entry point names the reference interface it
 * replaces (file:line under hvasbath/beat v2.0.5).  INTEGRATION.md shows the ctypes
 * stubs a BEAT maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative BEATAMD_E* code;
 *    beatamd_last_error() returns the thread-local message of the last failure.
 *  - all arrays are C-contiguous IEEE float64 unless stated; indices are int32.
 *  - array arguments may be HOST pointers (numpy) or DEVICE pointers (HBM, e.g. a torch
 *    tensor's data_ptr()); the library detects which (hipPointerGetAttributes) and
 *    stages host buffers through context-owned HBM scratch.  With device pointers no
 *    copy is made and the call is asynchronous on the context stream; with host
 *    pointers the call returns after the results are back in host memory.
 *  - "C" is the number of Markov chains in the batch: chains are the batch dimension
 *    of every kernel (the reference evaluates one chain per call).
 *  - one beatamd_ctx per (process, GPU); a context is not thread-safe.
 */
#ifndef BEAT_AMD_H
#define BEAT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEATAMD_OK 0
#define BEATAMD_EINVAL (-1)   /* bad argument (reference: AttributeError / ValueError)   */
#define BEATAMD_EHIP (-2)     /* HIP runtime failure                                       */
#define BEATAMD_EINDEX (-3)   /* index outside the GF library (reference: numpy IndexError) */
#define BEATAMD_ENOMEM (-4)
#define BEATAMD_ENAN (-5)     /* non-finite likelihood at stage 0 (metropolis.py:279-284)  */
#define BEATAMD_ENOTPSD (-6)  /* covariance not positive definite (numpy.linalg.LinAlgError) */
#define BEATAMD_EBADCOV (-7)  /* weighted sample covariance of the population contains Inf/NaN
                               * (reference: ValueError of SMC.calc_covariance, sampler/smc.py:167-186) */

/* ABI revision: bumped whenever an entry point changes its signature or an error code is added
 * (1.1: beatamd_weights_update gained kind/count in round 2; BEATAMD_EBADCOV).  beat_amd/_lib.py
 * refuses a library whose revision differs from the header it was written against. */
#define BEATAMD_VERSION 120

#define BEATAMD_NEAREST_NEIGHBOR 0 /* interpolation="nearest_neighbor" */
#define BEATAMD_MULTILINEAR 1      /* interpolation="multilinear"      */

#define BEATAMD_W_SCALAR 0 /* W_i = w_i * I   (chol_inverse of sigma^2 I)            */
#define BEATAMD_W_DENSE 1  /* W_i dense (M,M) row-major, upper-triangular content    */

typedef struct beatamd_ctx beatamd_ctx;

const char *beatamd_last_error(void);
int beatamd_version(void);

/* ---------------------------------------------------------------- context --------- */
int beatamd_ctx_create(int device, beatamd_ctx **out);
int beatamd_ctx_destroy(beatamd_ctx *ctx);
/* launch on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).
 * NULL is a valid handle: the HIP null (legacy default) stream, which is what torch uses
 * unless a side stream is current.  beatamd_ctx_use_own_stream goes back to the context's
 * private non-blocking stream. */
int beatamd_ctx_set_stream(beatamd_ctx *ctx, void *hip_stream);
int beatamd_ctx_use_own_stream(beatamd_ctx *ctx);
int beatamd_ctx_synchronize(beatamd_ctx *ctx);
/* The A/B and test knobs of the stacking path (environment variables BEATAMD_GF_*, BEATAMD_GS_*, BEATAMD_GC_*,
 * BEATAMD_GR_*, BEATAMD_WS_MAP, BEATAMD_SWEEP_V1; defaults = the shipped path) are read ONCE, when the context is
 * created.  This call reads them again.  A context created while BEATAMD_KNOBS_LIVE=1 is in the environment re-reads
 * them at every stacking call (test suites and A/B tools that compare kernels inside one process). */
int beatamd_ctx_reload_knobs(beatamd_ctx *ctx);
/* Device-resident step counter of the proposal generator.  With a counter set (uint32 in device
 * memory, NULL to unset) beatamd_proposal_draw / _univariate take the Philox step from it instead
 * of their `step` argument and add one to it afterwards: a Metropolis step captured in a HIP graph
 * (nothing in it synchronises with the host) then replays with fresh, reproducible draws --
 * the launch-bound geometry-mode problems run from a graph (beat_amd/sampler/metropolis.py). */
int beatamd_ctx_set_step_counter(beatamd_ctx *ctx, uint32_t *device_counter);
/* per-kernel HIP-event timing on the launch stream (bench.py roofline leg).
 * kernel names: "sweep", "tables", "gfstack", "quadform", "geostack", "finish", "astep" */
int beatamd_ctx_enable_timing(beatamd_ctx *ctx, int on);
int beatamd_ctx_kernel_time(beatamd_ctx *ctx, const char *kernel, double *total_ms,
                            int64_t *launches);
int beatamd_ctx_reset_timing(beatamd_ctx *ctx);
/* name and template arguments of the stacking kernel the most recent stack_all / logp / astep call
 * launched, e.g. "k_gfstack_dma<8,1,1,64,1>" (wavefronts per workgroup, rows per patch, epilogue,
 * samples per tile, ds_read_b64 layout) or "k_gfstack<0,1,1,2,0>" (streaming form). */
int beatamd_ctx_last_kernel(beatamd_ctx *ctx, char *buf, int64_t buflen);
/* chain-shared stacking kernels: distinct library rows per (chain group, target, patch) of the most
 * recent launch -- mean, maximum, and the bytes of library rows staged from HBM (sum x N x 8).
 * chains_per_group = 0 / row_bytes = 0 when the streaming kernel ran.  Synchronises. */
int beatamd_ctx_gf_group_stats(beatamd_ctx *ctx, int64_t *chains_per_group, double *mean_rows,
                               int64_t *max_rows, int64_t *row_bytes);
/* which stacking kernel the selection chose for the most recent launch and why, in words (chains per group, LDS row
 * slots, whether patches are staged in several row passes), and -- when asked for (non-NULL; synchronises) -- the row
 * passes per (chain group, target, patch) of that launch: 1 everywhere unless a patch touched more distinct library
 * rows than an LDS row buffer holds.  No reference counterpart: the reference gathers one chain's rows by fancy
 * indexing (beat/ffi/base.py:651-704); this reports how the batch shares them. */
int beatamd_ctx_gf_plan(beatamd_ctx *ctx, char *buf, int64_t buflen, double *mean_passes, int64_t *max_passes);
/* the chain-group size of the lane <-> chain stacking kernels is MEASURED on the first call of a problem shape (every
 * candidate launched twice on the real inputs, the fastest kept; from 1024 chains on only 512 / 256 are candidates):
 * this returns the most recent measurement in words ("group size for 4096 chains (...): 512: 35.8 ms, 256: 61.2 ms ->
 * 512"), empty before the first one.  BEATAMD_VERBOSE=1 prints the same line to stderr when it happens.  No reference
 * counterpart (the reference has no batch; beat/ffi/base.py:607-709 stacks one chain). */
int beatamd_ctx_gf_tune_log(beatamd_ctx *ctx, char *buf, int64_t buflen);
/* Libraries of short traces (nsamples <= 256: the usual 60 s at 2 Hz of an FFI set-up) are stacked in R patch
 * RANGES -- the same memory viewed as [T*R, P/R, D, S, N], the ranges' partial synthetics summed in range order -- so that
 * T * ceil(N/64) walks of P serial steps become R times as many, R times shorter ones for the device's compute units.
 * This is the rule (a pure function; num_cu <= 0: 256): R = the divisor of P with >= 32 patches per range that minimises
 * ceil(walks * R / num_cu) * (P / R + 5); 1 = the library as it is.  What a call did is in beatamd_ctx_gf_plan.  No reference
 * counterpart (beat/ffi/base.py:607-709 stacks one chain on one core). */
int32_t beatamd_gf_patch_ranges(int64_t ntargets, int64_t npatches, int64_t nsamples, int32_t num_cu);

/* how a batch of C chains is cut into its chain groups (scheduling only; results never depend on it): recursive
 * bisection of the batch along the key in which a part's chains spread wider -- the fused model path hands the hypocentre
 * (strike, dip) of every chain, so that a group covers a compact piece of the fault and stages fewer distinct library rows.
 *   key0 / key1 [C] (device)   members [ceil(C / chains_per_group) * chains_per_group] (host): members[g * cpg + i] =
 *   i-th chain of group g, 0xffffffff behind the last chain.  Batches beyond 8192 chains (or 64 groups) are cut chunk by chunk: chunks of min(8192 / chains_per_group, 64) whole groups as the chains come, each bisected on its own (chains_per_group <= 8192).
 * No reference counterpart (the reference evaluates one chain per process, beat/sampler/base.py:428-595). */
int beatamd_ctx_gf_chain_groups(beatamd_ctx *ctx, int64_t C, const double *key0, const double *key1,
                                int64_t chains_per_group, uint32_t *members);

/* ---------------------------------------------------------------- fast sweep -------
 * replaces: fast_sweep_ext.fast_sweep(slowness, patch_size, h_strk, h_dip, num_strk,
 *           num_dip)                       beat/fast_sweeping/fast_sweep_ext.c:120-245
 *           pytensorf.Sweeper.perform      beat/pytensorf.py:443-500
 * Batched over chains.  Same argument slots as the C extension: rows i in
 * [0,num_strk), columns j in [0,num_dip), flat index i*num_dip + j (BEAT passes
 * dip in the "strk" slots, pytensorf.py:473-482).
 *   slowness [C, num_strk*num_dip]   h_strk/h_dip [C] int32   out [C, num_strk*num_dip]
 * A hypocentre index outside the grid is BEATAMD_EINVAL (the reference writes out of
 * bounds). */
int beatamd_fast_sweep_batch(beatamd_ctx *ctx, const double *slowness, double patch_size,
                             const int32_t *h_strk, const int32_t *h_dip, int32_t num_strk,
                             int32_t num_dip, int64_t C, double *out);

/* ---------------------------------------------------------------- GF libraries -----
 * replaces: SeismicGFLibrary (5-D float64 array (ntargets, npatches, ndurations,
 *           nstarttimes, nsamples) + index maps)          beat/ffi/base.py:320-709
 *           init_optimization() / parallel.memshare: one HBM-resident copy per GPU
 *                                              beat/ffi/base.py:387-404, parallel.py:285-439 */
int beatamd_seis_gflib_create(beatamd_ctx *ctx, int64_t ntargets, int64_t npatches,
                              int64_t ndurations, int64_t nstarttimes, int64_t nsamples,
                              double starttime_min, double starttime_sampling,
                              double duration_min, double duration_sampling, int32_t *lib_id);
/* copy `count` doubles (host or device source) to element offset `offset` of the library */
int beatamd_seis_gflib_upload(beatamd_ctx *ctx, int32_t lib_id, const double *src,
                              int64_t offset, int64_t count);
/* use a caller-owned device allocation as the library storage (no copy) */
int beatamd_seis_gflib_adopt(beatamd_ctx *ctx, int32_t lib_id, double *device_ptr);
/* float storage (SURVEY 8(f) row 2 "optional fp32 layout"; the reference keeps its libraries in
 * float64, beat/ffi/base.py:381-385 tconfig.floatX aside).  beatamd_seis_gflib_round_to_f32 ROUNDS THE
 * LIBRARY IN PLACE -- the float64 storage (also a caller-owned adopted allocation) is overwritten with
 * (double)(float)value, irreversibly, values change by up to 6e-8 relative -- and keeps a float copy of
 * the same values in HBM, so that every kernel, whichever copy it reads, works on ONE library.
 * Accumulation, weights and likelihoods stay float64.  Never called by the library itself.
 * beatamd_ffi_model_set_f32 lets a wavemap of a model read the float copies where a kernel exists for
 * them (on = 0: the float64 kernels on the same rounded values; it does NOT restore anything).  An upload
 * into the library or an in-place whitening of its rows (beatamd_whiten_rows) drops the float copy and
 * takes every wavemap off it: the copies would no longer agree. */
int beatamd_seis_gflib_round_to_f32(beatamd_ctx *ctx, int32_t lib_id);
int beatamd_ffi_model_set_f32(beatamd_ctx *ctx, int32_t model_id, int32_t wavemap_index, int32_t on);
int beatamd_seis_gflib_device_ptr(beatamd_ctx *ctx, int32_t lib_id, double **device_ptr);
int beatamd_seis_gflib_destroy(beatamd_ctx *ctx, int32_t lib_id);

/* replaces: SeismicGFLibrary.stack_all(durations, starttimes, slips, targetidxs,
 *           patchidxs, interpolation)                      beat/ffi/base.py:607-709
 *           (incl. starttimes2idxs :486-521, durations2idxs :535-568)
 *   durations [C,P]  starttimes [C,T,P]  slips [C,P]  ->  out [C,T,N]
 * An index outside the library is BEATAMD_EINDEX (numpy IndexError); negative indices
 * wrap like numpy. */
int beatamd_seis_stack_all_batch(beatamd_ctx *ctx, int32_t lib_id, int64_t C,
                                 const double *durations, const double *starttimes,
                                 const double *slips, int32_t interpolation, double *out);

/* replaces: GeodeticGFLibrary.stack_all(slips) = G.T.dot(slips)  beat/ffi/base.py:292-305
 *   G [P,Nobs] (uploaded once)   slips [C,P]  ->  out [C,Nobs]; accumulate!=0 adds */
int beatamd_geo_gflib_create(beatamd_ctx *ctx, int64_t npatches, int64_t nobs, const double *G,
                             int32_t *lib_id);
int beatamd_geo_gflib_destroy(beatamd_ctx *ctx, int32_t lib_id);
int beatamd_geo_stack_all_batch(beatamd_ctx *ctx, int32_t lib_id, int64_t C,
                                const double *slips, int32_t accumulate, double *out);

/* ---------------------------------------------------------------- likelihood -------
 * replaces: multivariate_normal_chol(datasets, weights, hyperparams, residuals)
 *                                              beat/models/distributions.py:72-140
 * A "weight set" holds what the reference keeps in pytensor shared variables per
 * dataset: W_i = Covariance.chol_inverse (heart.py:211-237), slog_pdet_i
 * (heart.py:239-253) and M_i = nsamples.  All datasets of a set share M.
 *   kind SCALAR: weights [nd]      kind DENSE: weights [nd, M, M]
 * update = SeismicComposite.update_weights (seismic.py:1509-1534): same call again. */
int beatamd_weights_create(beatamd_ctx *ctx, int32_t kind, int64_t ndatasets, int64_t M,
                           const double *weights, const double *slog_pdet, int32_t *wset_id);
/* kind and count (number of doubles in `weights`) are validated against the set: a wavemap whose
 * library was pre-whitened holds scalar weights and rejects a dense update (re-whiten instead). */
int beatamd_weights_update(beatamd_ctx *ctx, int32_t wset_id, int32_t kind, int64_t count,
                           const double *weights, const double *slog_pdet);
int beatamd_weights_destroy(beatamd_ctx *ctx, int32_t wset_id);
/* BANDED whitening operators.  The reference's "exponential" noise structure (beat/covariance.py:24-51:
 * C_ij = exp(-|i-j| dt / t0), a Markov kernel) has a BIDIAGONAL W = chol(inv(C)).T -- what numpy's inv + cholesky leave
 * outside the band is rounding residue (~2e-15 of the largest entry).  A DENSE weight set whose matrices are all
 * upper-triangular with no entry further than 16 columns right of the diagonal above 2^-40 of its matrix's largest entry is
 * therefore evaluated on its band (two products per sample for the bidiagonal case instead of a matrix row; the dropped
 * entries change a whitened sample by < M * 2^-40 of its largest term, far inside the 1e-6 tolerance of the path).
 * *band = the half bandwidth in use, -1 when the dense kernel is used.  BEATAMD_QF_BAND=0 (environment) keeps the dense
 * kernel for every set. */
int beatamd_weights_band(beatamd_ctx *ctx, int32_t wset_id, int64_t *band);
/* the same with what the banded evaluation leaves out: max_dropped_rel = the largest |entry| beyond the band relative to the
 * largest |entry| of its own row (<= 2^-40 by construction; 0 for an exactly banded operator; rounding residue of
 * inv + cholesky, ~2e-15, for the reference's "exponential" structure, beat/covariance.py:24-51 through heart.py:201-253).
 * A caller logs it once per weights_create / weights_update (beat_amd.models.problem does); BEATAMD_VERBOSE=1 prints the
 * detection to stderr.  band = -1: evaluated by the dense kernel. */
int beatamd_weights_band_info(beatamd_ctx *ctx, int32_t wset_id, int64_t *band, double *max_dropped_rel);
/*   residuals [C, nd, M]   hp [C, nd] (hyperparameter already resolved per dataset,
 *   distributions.py:117-126)   ->   logpts [C, nd]                                  */
int beatamd_mvn_chol_logp_batch(beatamd_ctx *ctx, int32_t wset_id, int64_t C,
                                const double *residuals, const double *hp, double *logpts);

/* replaces: LaplacianDistributerComposite.get_formula / _eval_prior
 *                                              beat/models/laplacian.py:88-139
 *   L [P,P]  slips [C, nvar, P]  hp [C]  ->  out [C]  (sum over slip variables)      */
int beatamd_laplacian_create(beatamd_ctx *ctx, int64_t npatches, const double *L,
                             double logdet, int32_t *lap_id);
int beatamd_laplacian_destroy(beatamd_ctx *ctx, int32_t lap_id);
int beatamd_laplacian_logp_batch(beatamd_ctx *ctx, int32_t lap_id, int64_t C, int64_t nvar,
                                 const double *slips, const double *hp, double *out);

/* ---------------------------------------------------------------- fused FFI model ---
 * replaces: the compiled logp_forw_func(q) of a DistributionOptimizer (FFI) problem:
 *           sampler/base.py:598-615 logp_forw; graph built by
 *           SeismicDistributerComposite.get_formula   beat/models/seismic.py:1210-1349
 *           GeodeticDistributerComposite.get_formula  beat/models/geodetic.py:1030-1084
 *           LaplacianDistributerComposite.get_formula beat/models/laplacian.py:98-139
 *           Problem.built_model (like = sum)          beat/models/problems.py:212-248
 * Q [C, nparams] -> LL [C, nllk]; nllk = sum(seismic datasets) + sum(geodetic
 * datasets) + (laplacian ? 1 : 0) + 1, ordered seis_like.., geo_like.., laplacian_like,
 * like  (SURVEY Appendix C "Outputs").
 * Offsets are positions in the flat parameter vector q (DictToArrayBijection order,
 * taken from the host model); -1 = variable absent / fixed. */
typedef struct beatamd_ffi_layout {
    int64_t nparams;
    int32_t nvar;            /* slip variables (uparr, uperp, ...)                        */
    int64_t slip_off[4];     /* [nvar] offset of each slip variable (size npatches)        */
    int64_t durations_off;   /* size npatches                                              */
    int64_t velocities_off;  /* size npatches                                              */
    int64_t nuc_strike_off;  /* size nsubfaults                                            */
    int64_t nuc_dip_off;     /* size nsubfaults                                            */
    int64_t time_off;        /* size nsubfaults                                            */
    int64_t h_laplacian_off; /* size 1, -1 if no laplacian                                 */
} beatamd_ffi_layout;

int beatamd_ffi_model_create(beatamd_ctx *ctx, const beatamd_ffi_layout *layout,
                             int32_t nsubfaults, const int32_t *n_patch_dip,
                             const int32_t *n_patch_strike, const double *patch_size,
                             int32_t *model_id);
/* one seismic wavemap (seismic.py:1274-1341):
 *   lib_ids [nvar]   data [T,N]   wset_id over the T datasets
 *   hp_off [T]: offset in q of each dataset's hyperparameter
 *   shift_off [T]: offset in q of each target's station time-shift, or NULL
 *                  (hierarchicals[time_shifts_id][station_correction_idxs])           */
int beatamd_ffi_model_add_wavemap(beatamd_ctx *ctx, int32_t model_id, const int32_t *lib_ids,
                                  const double *data, int32_t wset_id, const int64_t *hp_off,
                                  const int64_t *shift_off, int32_t interpolation);
/* geodetic composite (geodetic.py:1065-1081): geo lib per slip var, data [Nobs],
 * odws [Nobs], dataset sizes [nd] (srmap), one wset per dataset (sizes differ) */
int beatamd_ffi_model_add_geodetic(beatamd_ctx *ctx, int32_t model_id, const int32_t *geo_lib_ids,
                                   const double *data, const double *odws, int32_t ndatasets,
                                   const int64_t *dataset_sizes, const int32_t *wset_ids,
                                   const int64_t *hp_off);
/* geometry-mode geodetic composite (GeodeticGeometryComposite.get_formula,
 * beat/models/geodetic.py:605-659): the displacements come from analytic half-space sources
 * instead of a linear GF library -- rectangular dislocations (Okada 1985; kind 0) and Mogi
 * point sources (kind 1) -- then LOS projection (:642), residual * odw and
 * multivariate_normal_chol exactly as above.  The reference obtains the displacements from
 * pyrocko's GF-store engine (heart.geo_synthetics, heart.py:4158-4239; not in its tree):
 * parity with BEAT is unpinned for this entry, it is pinned to Okada's published check values.
 *   per source 10 parameters: east_shift north_shift depth [km] (centre of the top edge),
 *   strike dip rake [deg], length width [km], slip [m] (Mogi: volume change [m^3]),
 *   opening_fraction;  param_off[nsrc*10] = offset in q or -1 -> param_fixed value
 *   east/north [Nobs] observation points [km], los [Nobs,3] = (Sn, Se, Su) (heart.py:1381-1410) */
int beatamd_ffi_model_add_geodetic_geometry(beatamd_ctx *ctx, int32_t model_id, int32_t nsrc,
                                            const int32_t *kind, const int64_t *param_off,
                                            const double *param_fixed, int64_t nobs,
                                            const double *east, const double *north,
                                            const double *los, double nu, const double *data,
                                            const double *odws, int32_t ndatasets,
                                            const int64_t *dataset_sizes, const int32_t *wset_ids,
                                            const int64_t *hp_off);
int beatamd_ffi_model_set_laplacian(beatamd_ctx *ctx, int32_t model_id, int32_t lap_id);
int beatamd_ffi_model_nllk(beatamd_ctx *ctx, int32_t model_id, int64_t *nllk);
int beatamd_ffi_model_destroy(beatamd_ctx *ctx, int32_t model_id);

/* replaces: SeismicComposite.get_synthetics(point) for the distributed-slip composite
 *           (sweep -> start times -> stack_all), beat/models/seismic.py:1351-1507, as
 *           update_weights / analyse_noise call it at the MAP point of a stage (:1509-1534,
 *           beat/covariance.py:333-395)
 *   Q [C,nparams] -> out [C,T,N] of wavemap `wavemap_index`: synthetics, or (residuals != 0)
 *   data - synthetics (seismic.py:1332).  An index outside the library is BEATAMD_EINDEX; the rows of
 *   such a chain are then UNSPECIFIED (they differ between the stacking kernels; the reference raises
 *   IndexError before producing any). */
int beatamd_ffi_synthetics_batch(beatamd_ctx *ctx, int32_t model_id, int32_t wavemap_index, int64_t C,
                                 const double *Q, int32_t residuals, double *out);

/* logp_forw_func batched: Q [C,nparams] -> LL [C,nllk] */
int beatamd_ffi_logp_batch(beatamd_ctx *ctx, int32_t model_id, int64_t C, const double *Q,
                           double *LL);

/* replaces: Metropolis.astep for C chains at once   beat/sampler/metropolis.py:276-422
 *           (stage > 0, check_bound=True, continuous variables)
 *   Q0 [C,nparams] current points (updated in place)
 *   L0 [C,nllk]    current likelihood vectors = chain_previous_lpoint (updated in place)
 *   delta [C,nparams] proposal_samples_array[stage_sample] rows
 *   scaling [C]    per-chain step scaling;  lower/upper [nparams] Uniform prior boxes
 *   log_u [C]      log of the MH uniforms (metrop_select)
 *   beta           tempering parameter;  accepted [C] int32 out (1 = moved)            */
int beatamd_ffi_astep_batch(beatamd_ctx *ctx, int32_t model_id, int64_t C, double *Q0,
                            double *L0, const double *delta, const double *scaling,
                            const double *lower, const double *upper, const double *log_u,
                            double beta, int32_t *accepted);
/* same with one beta per chain: the replicas of a parallel-tempering ladder
 * (worker_process / sample_pt_chain, beat/sampler/pt.py:651-790) advance in one batch */
int beatamd_ffi_astep_batch_betas(beatamd_ctx *ctx, int32_t model_id, int64_t C, double *Q0,
                                  double *L0, const double *delta, const double *scaling,
                                  const double *lower, const double *upper, const double *log_u,
                                  const double *betas, int32_t *accepted);

/* The whole Metropolis step with the proposal drawn on the device: what the two calls
 * beatamd_proposal_draw[_univariate] + beatamd_ffi_astep_batch[_betas] do, as one call that touches
 * the host for nothing (the sampling loop of a stage is then one C call per step, capturable in a
 * HIP graph together with the device-resident step counter, beatamd_ctx_set_step_counter).
 * replaces: Metropolis.astep incl. its proposal draw and bookkeeping
 *           beat/sampler/metropolis.py:276-422 (proposal_dist(n_steps) :289-292, accepted :386-410)
 *   factor, K, kind, df   kind -1: multivariate proposal, factor [K,nparams], delta = z.factor
 *                         (df > 0: multivariate-t row scale, base.py:35-71);
 *                         kind 0/1/2: per-parameter Normal/Cauchy/Laplace, factor = scales [nparams]
 *   seed, step, first_chain   the Philox key/counter of beatamd_proposal_draw (same numbers)
 *   beta / betas          betas != NULL: one beta per chain (parallel tempering), else the scalar
 *   accepted [C] int32 out; accepted_sum [C] int32 (+= accepted) and n_accepted [1] int64
 *   (+= number of moves) may be NULL.  Q0, L0 and the counters are device pointers.
 * For nparams, K <= 64 the draws, the factor product and the prior-box test are one launch.       */
int beatamd_ffi_mstep_batch(beatamd_ctx *ctx, int32_t model_id, int64_t C, double *Q0, double *L0,
                            const double *factor, int64_t K, int32_t kind, int32_t df, uint64_t seed,
                            uint32_t step, int64_t first_chain, const double *scaling,
                            const double *lower, const double *upper, double beta, const double *betas,
                            int32_t *accepted, int32_t *accepted_sum, int64_t *n_accepted);

/* ---------------------------------------------------------------- noise covariance -------
 * Per-stage re-estimation of the data covariances (update_weights with the "non-toeplitz"
 * structure, seismic.py:1509-1534 -> covariance.py:397-427): the reference runs O(n^2) Python
 * loops per dataset.
 * replaces: covariance.autocovariance(data)            beat/covariance.py:716-736
 *   data [nd,n], mean [nd] (= data.mean(), computed by the caller) -> out [nd,n]
 *   same term order as the reference loop: bitwise equal results                             */
int beatamd_autocovariance_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *data,
                                 const double *mean, double *out);
/* replaces: toeplitz(coeffs) * stds[:,None] * stds[None,:]   beat/covariance.py:739-771
 *   coeffs [nd,n], stds [nd,n] -> out [nd,n,n]                                               */
int beatamd_scaled_toeplitz_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *coeffs,
                                  const double *stds, double *out);

/* ---------------------------------------------------------------- SMC stage transition ----
 * The population (end points Q [C,nparams], likelihood vectors L [C,nllk]) stays in HBM between
 * stages; these entries run on device pointers (host pointers are staged like everywhere else).
 * All reductions have a fixed order, so every rank of a multi-GPU run that holds the same gathered
 * arrays computes bit-identical decisions.
 *
 * replaces: SMC.calc_beta                                   beat/sampler/smc.py:133-165
 *   likelihoods: C values `stride` doubles apart (the `like` column of L: stride = nllk)
 *   -> *beta_new (host), weights [C] (importance weights of the last bisection midpoint,
 *      normalised).  exp() is the device's (<= 1 ulp from glibc/numpy): weights agree with
 *      the reference to ~1e-15 relative, beta exactly unless a bisection decision sits on a tie. */
int beatamd_smc_calc_beta(beatamd_ctx *ctx, int64_t C, const double *likelihoods, int64_t stride,
                          double beta, double coef_variation, double *beta_new, double *weights);
/* final stage (smc.py:526-529): weights = exp(dbeta (l - max l)) / sum */
int beatamd_smc_stage_weights(beatamd_ctx *ctx, int64_t C, const double *likelihoods, int64_t stride,
                              double dbeta, double *weights);
/* replaces: SMC.resample (Kitagawa's deterministic resampling)   beat/sampler/smc.py:290-324
 *   weights [C], aux = the single uniform draw (host RNG, shared by all ranks)
 *   -> indexes [C] int32 = np.repeat(parents, N_childs); sequential cumulative sum as np.cumsum:
 *      bit-exact indices */
int beatamd_smc_resample(beatamd_ctx *ctx, int64_t C, const double *weights, double aux,
                         int32_t *indexes);
/* replaces: SMC.calc_covariance + MultivariateNormalProposal   smc.py:167-186, base.py:163-167
 *   factor [C,nparams] with factor^T factor = np.cov(population, aweights=weights, bias=False):
 *   rows z . factor (z ~ N(0, I_C)) are draws of N(0, cov) without factoring the covariance */
int beatamd_smc_population_factor(beatamd_ctx *ctx, int64_t C, int64_t nparams,
                                  const double *population, const double *weights, double *factor);
/* replaces: proposal_dist(n_steps) rows + the Metropolis uniforms   metropolis.py:289-292, 355-358
 *           multivariate_t_rvs                                       base.py:35-71
 *   delta [C,nparams] = z [C,K] . factor [K,nparams] (FP64 MFMA), z from Philox4x32-10 keyed by
 *   `seed` with counter (pair, first_chain + c, step, stream): a chain's draws do not depend on
 *   the sharding.  df = 0: multivariate normal; df > 0: multivariate t (1 = Cauchy): rows divided
 *   by sqrt(chi2(df) / df).  log_u [C] (nullable) = log of U(0,1). */
int beatamd_proposal_draw(beatamd_ctx *ctx, int64_t C, int64_t K, int64_t nparams,
                          const double *factor, uint64_t seed, uint32_t step, int64_t first_chain,
                          int32_t df, double *delta, double *log_u);
/* replaces: NormalProposal / CauchyProposal / LaplaceProposal / PoissonProposal (per-parameter families)
 *                                                              beat/sampler/base.py:129-160
 *   delta [C,nparams]: every component an independent draw times scale[j] (the reference's
 *   Metropolis passes scale = ones, metropolis.py:209-212); same Philox counters as above
 *   (streams 3, 4); log_u [C] (nullable) = log of U(0,1).  Poisson: poisson(lam = scale[j]) - scale[j]
 *   (inversion of one uniform by sequential search; scale[j] <= 500, NaN rows beyond). */
#define BEATAMD_PROPOSAL_NORMAL 0
#define BEATAMD_PROPOSAL_CAUCHY 1
#define BEATAMD_PROPOSAL_LAPLACE 2
#define BEATAMD_PROPOSAL_POISSON 3   /* poisson(lam = scale) - scale (beat/sampler/base.py:150-155) */
int beatamd_proposal_draw_univariate(beatamd_ctx *ctx, int64_t C, int64_t nparams, int32_t kind,
                                     const double *scale, uint64_t seed, uint32_t step, int64_t first_chain,
                                     double *delta, double *log_u);
/* out[i,:] = src[indexes[i],:]: chains restart at their resampled parents (sampler/base.py:541-571),
 * replica exchange permutation (pt.py:573-633).  An index outside [0,nrows_src) is BEATAMD_EINDEX
 * at the next synchronisation. */
int beatamd_gather_rows(beatamd_ctx *ctx, int64_t nout, int64_t ncols, const double *src,
                        int64_t nrows_src, const int32_t *indexes, double *out);
/* ---- a Metropolis step in pieces, for models with a collective between forward model and acceptance (libraries sharded
 * by TARGET over ranks; beat_amd/models/sharded.py).  All pointers are device pointers.
 *
 * beatamd_like_assemble: the full likelihood vectors LL [C, nllk] (layout of the unsharded model: datasets..., like) from
 *   the all-gathered block `gathered` [nsrc, C] -- row r goes to column dst_col[r] (host int32 [nsrc]); dst_col[r] = -1
 *   marks a rank's FLAG row (NaN in it = that rank saw the chain's start times / durations leave the library grid: the
 *   reference raises IndexError there, beat/ffi/base.py:486-568) -- + the replicated columns (geodetic datasets,
 *   Laplacian) local_ll[c*local_ld + local_col0 .. +n_rest) -> columns rest_dst0..; then like = the composites' sums in
 *   order, summed (beat/models/problems.py:227-247; group_end [ngroups] host: exclusive column ends), NaN for flagged chains.
 * beatamd_metropolis_propose: q = q0 + delta * scaling, prior box test, out-of-box rows parked on q0 (metropolis.py:313-343).
 * beatamd_metropolis_accept: tempered acceptance iff in bounds, isfinite(mr), log u < mr with mr = beta (like' - like)
 *   (betas [C] non-NULL: per chain) -- metropolis.py:344-385 + pymc metrop_select; Q0 / L0 updated in place. */
int beatamd_like_assemble(beatamd_ctx *ctx, int64_t C, int64_t nllk, int64_t nsrc, const double *gathered,
                          const int32_t *dst_col, const double *local_ll, int64_t local_ld, int64_t local_col0,
                          int64_t n_rest, int64_t rest_dst0, int32_t ngroups, const int32_t *group_end, double *LL);
int beatamd_metropolis_propose(beatamd_ctx *ctx, int64_t C, int64_t nparams, const double *Q0, const double *delta,
                               const double *scaling, const double *lower, const double *upper, double *Qprop,
                               int32_t *inbounds);
int beatamd_metropolis_accept(beatamd_ctx *ctx, int64_t C, int64_t nparams, int64_t nllk, double *Q0, double *L0,
                              const double *Qprop, double *Lprop, const int32_t *inbounds, const double *log_u, double beta,
                              const double *betas, int32_t *accepted);
/* replaces: the per-chain step-size tuning of Metropolis.astep   metropolis.py:294-306
 *   scaling [C] *= pymc's tune factor of accepted[c] / tune_interval; accepted [C] reset to 0 */
int beatamd_metropolis_tune(beatamd_ctx *ctx, int64_t C, double *scaling, int32_t *accepted,
                            int32_t tune_interval);

/* ---------------------------------------------------------------- library whitening ------
 * rows [nrows, N] (device, in place) <- rows . W^T, W [N,N] = chol_inverse of one dataset: the
 * dense W.r of multivariate_normal_chol (distributions.py:128) applied once to every library row
 * of the dataset instead of once per chain step (SeismicWavemap.prewhitened).  FP64 MFMA GEMM;
 * an upper-triangular W (heart.py:233) skips its zero half. */
int beatamd_whiten_rows(beatamd_ctx *ctx, double *rows, int64_t nrows, int64_t N, const double *W);
/* the same for all datasets of a wavemap in one call: rows [nbatch, nrows, N] (device, in place),
 * W [nbatch, N, N] (host or device).  Upper-triangular operators are applied column block by column block
 * in ascending order with no second buffer (a product column n needs the row's entries k >= n only);
 * anything else goes dataset by dataset through beatamd_whiten_rows.  This is what a covariance update of
 * a pre-whitened model costs per stage (seismic.py:1509-1534 on 62.9 GB of rows at config 3). */
int beatamd_whiten_rows_batch(beatamd_ctx *ctx, double *rows, int64_t nbatch, int64_t nrows, int64_t N,
                              const double *W);

/* ---------------------------------------------------------------- whitening operator ------
 * replaces: heart.Covariance.chol_inverse / .log_pdet                beat/heart.py:216-253
 *           (numpy.linalg.inv + cholesky, + slogdet) for a stack of nd covariance matrices, as
 *           update_weights needs them per stage for every dataset (seismic.py:1509-1534)
 *   covs [nd,n,n] symmetric positive definite -> W [nd,n,n] = cholesky(inv(C)).T (upper
 *   triangular), log_pdet [nd] = log det C.  Blocked factorisation of the exchange-flipped matrix
 *   on the FP64 matrix cores (no explicit inverse of C).  A matrix that is not positive definite
 *   is BEATAMD_ENOTPSD (numpy raises LinAlgError). */
int beatamd_chol_inverse_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *covs, double *W,
                               double *log_pdet);
/* same, for callers that repair the failing matrices themselves (utility.ensure_cov_psd /
 * repair_covariance, beat/utility.py:1034-1138; get_data_covariances, beat/covariance.py:413-427):
 * not_psd [nd] = 1 where the factorisation met a non-positive pivot (W / log_pdet of that matrix
 * are meaningless), no error is raised. */
int beatamd_chol_inverse_batch_flags(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *covs, double *W,
                                     double *log_pdet, int32_t *not_psd);

/* replaces: the Cholesky factor inside pymc's MultivariateNormalProposal (base.py:163-186,
 *           numpy.linalg.cholesky of the proposal covariance) for the stage proposals of SMC:
 *   factor [K,n] (beatamd_smc_population_factor, K = number of chains) -> R [n,n] upper triangular
 *   with R^T R = factor^T factor, so that z[n] . R has the distribution of z[K] . factor.  Used when
 *   the population is larger than the number of parameters (n normals per proposal row instead
 *   of K).  A Gram matrix that is not numerically positive definite is BEATAMD_ENOTPSD: the caller
 *   keeps the tall factor (a collapsed or too small population). */
int beatamd_factor_compact(beatamd_ctx *ctx, int64_t K, int64_t n, const double *factor, double *R);

/* replaces: nothing in the reference (it keeps W and multiplies per step); companion of
 *           beatamd_whiten_rows for update_weights (seismic.py:1509-1534) on a pre-whitened library:
 *   M [nd,n,n] = W_new . inv(W_old) for upper-triangular whitening operators, so that
 *   rows . W_new^T = (rows . W_old^T) . M^T -- the library and the data follow a covariance update in
 *   place.  A singular W_old is BEATAMD_ENOTPSD. */
int beatamd_whitening_ratio_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *W_new,
                                  const double *W_old, double *M);
/* replaces: nothing in the reference; the covariance update (covariance.py:307-325 estimates the noise on
 *           d - s) on a pre-whitened model, whose residual traces are W_old (d - s):
 *   X [nd,n] <- inv(W_t) . X[t] for upper-triangular W [nd,n,n] (back substitution, one trace per
 *   dataset; in place, host or device pointers).  A zero on a diagonal is BEATAMD_ENOTPSD. */
int beatamd_unwhiten_traces(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *W, double *X);
/* replaces: the data half of update_weights on a pre-whitened wavemap: new observed data
 *   [T,N] (already whitened) of wavemap `wavemap_index` of a compiled model */
int beatamd_ffi_model_update_data(beatamd_ctx *ctx, int32_t model_id, int32_t wavemap_index, const double *data);

/* ---------------------------------------------------------------- half-space synthetics ---
 * replaces: heart.geo_synthetics(engine, targets, sources, outmode)   beat/heart.py:4158-4239
 *           (what pytensorf.GeoSynthesizer.perform calls, pytensorf.py:88-123) for a HOMOGENEOUS
 *           HALF SPACE: rectangular dislocations (Okada 1985; kind 0) and Mogi sources (kind 1).
 *           The reference computes these displacements with pyrocko's layered GF-store engine
 *           (not in its tree): parity with BEAT is unpinned for this entry, it is pinned to Okada's
 *           published check values.
 *   params [C, nsrc, 10]: east_shift north_shift depth [km] (top-edge centre), strike dip rake
 *   [deg], length width [km], slip [m] (Mogi: volume change [m^3]), opening_fraction
 *   east/north [nobs] observation points [km]
 *   -> out [C, nsrc, nobs, 3] = (north, east, up) displacement [m] of every source at every point
 *      (the reference's per-(source, target) arrays `[n, e, -d]`, heart.py:4218-4224) */
int beatamd_halfspace_displacements_batch(beatamd_ctx *ctx, int64_t C, int32_t nsrc,
                                          const int32_t *kind, const double *params, int64_t nobs,
                                          const double *east, const double *north, double nu,
                                          double *out);

#ifdef __cplusplus
}
#endif
#endif /* BEAT_AMD_H */
