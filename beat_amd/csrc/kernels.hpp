// kernels.hpp -- launch interfaces of the HIP kernels (internal).
#pragma once
#include <algorithm>

#include "ctx.hpp"

namespace beatamd {

// ---- sweep.hip -------------------------------------------------------------------
int launch_sweep_explicit(beatamd_ctx *ctx, const double *slow, double h, const int32_t *hi,
                          const int32_t *hj, int ni, int nj, int64_t C, double *out);
int launch_sweep_model(beatamd_ctx *ctx, const FfiModel &m, const double *Q, int64_t C,
                       double *starttimes0, int32_t *chain_bad);

// ---- gfstack.hip -----------------------------------------------------------------
// Where the per-(chain,target,patch) start times come from.
struct StartTimeSrc {
    // explicit: st[(c*T + t)*P + p]
    const double *explicit_st = nullptr;
    // model: st = starttimes0[c*P + p] - Q[c*nparams + shift_off[t]]   (seismic.py:1283-1296)
    const double *starttimes0 = nullptr;
    const double *Q = nullptr;
    int64_t nparams = 0;
    const int64_t *shift_off = nullptr;  // device [T] or nullptr (no station corrections)
    // table slots: targets with the same shift variable share their index tables (nslot > 0: tslot [T] -> slot,
    // slot_shift_off [nslot]; device pointers)
    int32_t nslot = 0;
    const int32_t *tslot = nullptr;
    const int64_t *slot_shift_off = nullptr;
    // model mode: chains whose times fall outside the library grid are marked here (their
    // `like` becomes NaN, which the Metropolis step rejects); nullptr in the explicit API
    int32_t *chain_bad = nullptr;
};

// a strided view of per-chain vectors: value(c, k) = base[c*stride + off + k]
struct ChainVec {
    const double *base = nullptr;
    int64_t stride = 0, off = 0;
};

// Workgroup id -> work item for table kernels whose NEIGHBOURING items read the same cache lines (the [C,T,P,*] index
// tables: the entries of consecutive patches of a chain share a 128-byte line).  The hardware deals consecutive workgroup
// ids round-robin to the 8 XCDs, each with its own L2 -- eight neighbours would fetch the line eight times.  Inside every
// aligned block of 64 ids the items are dealt so that an XCD gets 8 CONSECUTIVE items (ids b = x mod 8 -> items 8x .. 8x+7).
#ifdef __HIPCC__
__device__ __forceinline__ int64_t xcd_items8(int64_t b, int64_t n)
{
    if (b >= (n & ~(int64_t)63)) return b;
    const int64_t r = b & 63;
    return (b & ~(int64_t)63) + (r & 7) * 8 + (r >> 3);
}
#endif

// patch ranges a short-trace library is stacked in (gfstack.hip; 1: as it is)
int gf_patch_ranges(int64_t T, int64_t P, int64_t N, int num_cu);

enum GfMode : int {
    GF_STORE_SYN = 0,     // out[c,t,n] = synthetics                      (stack_all)
    GF_RESID_SCALAR = 1,  // partial[c,t,tile] = sum (w_t (d - syn))^2     (fused logp, W = w I)
    GF_RESID_STORE = 2,   // out[c,t,n] = d[t,n] - synthetics              (feeds the dense W quadform)
    // bidiagonal whitening operator (the reference's "exponential" noise structure, covariance.py:24-51; band detected at
    // weights_create, quadform.hip): quad[c,t] = sum_i (w0_i r_i + w1_i r_{i+1})^2 (distributions.py:119-138 with a
    // bidiagonal W) without storing the residuals.  Kernels WITH this epilogue (k_gfstack_ws): the inner samples of a tile
    // in the kernel, the last sample of every tile -- its neighbour is the next tile's first residual -- by the tile-sum
    // kernel from two edge residuals per (chain, target, tile).  Kernels without it store the residuals (mode 2) and
    // launch_gfstack runs k_quadform_banded behind them: either way the caller gets `quad`.
    GF_RESID_BAND1 = 3
};

struct GfStackCall {
    const SeisLib *libs[4] = {nullptr, nullptr, nullptr, nullptr};
    int nvar = 1;
    ChainVec slips[4];
    ChainVec durations;
    StartTimeSrc st;
    int interp = 0;
    int64_t C = 0;
    int mode = GF_STORE_SYN;
    const double *data = nullptr;     // [T,N]   (modes 1,2)
    const double *wscalar = nullptr;  // [T]     (mode 1)
    double *out = nullptr;            // [C,T,N] (modes 0,2)
    double *quad = nullptr;           // [C,T]   (modes 1, 3) sum over tiles, fixed order
    const double *band_w = nullptr;   // [T,N,2] (mode 3) row i of the bidiagonal operator: (W[i,i], W[i,i+1]), 0 behind the end
    bool f32 = false;                 // rows from the libraries' float copies where the kernel supports it
    // optional scheduling hint: two per-chain sort keys that put chains which rupture alike next to each other
    // (the fused model path: hypocentre strike / dip of the first subfault).  Never changes a result.
    ChainVec order_key[2];
    // PATCH SPLIT (round 6, small-N libraries): the libraries are VIEWS [T*R, P/R, D, S, N] of the real ones (virtual
    // target t*R + r = target t, patches [r*P/R, (r+1)*P/R)), the tables are built per virtual slot and the stacking
    // kernels run unchanged on R times as many, R times shorter (target, tile) walks; slips / start times / durations
    // are read at the REAL patch r*P/R + p.  Set by launch_gfstack only.
    int patch_split = 1;
    const GfKnobs *knobs = nullptr;   // set by launch_gfstack (gf_knobs(ctx)): the selection functions read them here
    const int32_t *tslot = nullptr;   // set by launch_gfstack: table slot of a target when targets share tables (device [T])
};
int launch_gfstack(beatamd_ctx *ctx, const GfStackCall &call);
int launch_sum_tiles(beatamd_ctx *ctx, const double *partial, int64_t n, int ntile, double *quad,
                     const int *guard = nullptr, int want = 0);
// mode 3: quad[c,t] = sum over tiles of (partial + the tile's last sample: (w0 r_last + w1 r_first(next tile))^2), fixed order;
// edges [C*T, ntile, 2] = (first, last residual of the tile), NT samples per tile
int launch_sum_tiles_band1(beatamd_ctx *ctx, const double *partial, const double *edges, const double *band_w, int64_t C,
                           int64_t T, int64_t N, int ntile, int NT, double *quad, const int *guard = nullptr, int want = 0);
// g[i] = (double)(float)g[i]; g32[i] = (float)g[i]  (float-storage copy of a GF library)
int launch_round_to_f32(beatamd_ctx *ctx, double *g, float *g32, int64_t n);
// gfshared.hip: chain-shared variant (distinct rows staged once per chain group)
bool gfstack_shared_applicable(const GfStackCall &call, int *cg, int *ucap);
int gfstack_shared_candidates(const GfStackCall &call, int *cgs, int *ucaps);
int launch_gfstack_shared(beatamd_ctx *ctx, const GfStackCall &call, const uint32_t *rowoff,
                          const double *fac, int CG, int ucap, int64_t Ttab);

// gfcell.hip: multilinear stacking with the rows of a cell in registers (518-chain groups, row passes): k_gfstack_runs.
// *ovf (device, nullable on return): nonzero after the launch = the tables overflowed and nothing was stacked -- the
// caller enqueues k_gfstack behind it as a stand-in guarded by the same flag
// the chains of a batch cut into ngroups groups of cg chain slots by recursive bisection along the key in which a part's
// chains spread wider (k_gc_cut): members[g * cg + i] = i-th chain of group g, ~0 behind the last chain (one workgroup
// sorting in LDS: C <= 8192, <= 64 groups); members = nullptr when the batch is larger or a key is missing
int launch_chain_members(beatamd_ctx *ctx, int64_t C, const ChainVec key[2], int64_t cg, int64_t ngroups, const uint32_t **members,
                         int strips = 0);
bool gfstack_ml_applicable(const GfStackCall &call);
int launch_gfstack_ml(beatamd_ctx *ctx, const GfStackCall &call, const uint32_t *rowoff,
                      const double *fac, int64_t Ttab, const int **ovf);

// ---- quadform.hip ----------------------------------------------------------------
// quad[c,d] = || A_d x_{c,d} ||^2 ; A [nd or 1, M, M] row-major ; x(c,d,k) = X[c*xs_c + d*xs_d + k]
struct QuadformCall {
    const double *A = nullptr;
    int64_t a_stride = 0;  // elements between consecutive A_d (0: one shared A)
    int64_t M = 0, nd = 0, C = 0;
    const double *X = nullptr;
    int64_t xs_c = 0, xs_d = 0;
    int upper_tri = 0;
    double *quad = nullptr;  // [C, nd] with row stride q_stride
    int64_t q_stride = 0;
};
int launch_quadform(beatamd_ctx *ctx, const QuadformCall &call);
// banded upper-triangular operators (quadform.hip): the half bandwidth of a stack of matrices (entries beyond it are at most
// 2^-40 of their matrix's largest; scratch: nd * 8 + 8 bytes), the compact band [nd, M, band + 1], and the quadratic form on it
constexpr int QF_BAND_LIMIT = 16;
int launch_band_detect(beatamd_ctx *ctx, const double *A, int64_t nd, int64_t M, void *scratch, int64_t *band_host,
                       double *dropped_rel_host);
int launch_band_pack(beatamd_ctx *ctx, const double *A, int64_t nd, int64_t M, int64_t band, double *wb);
int launch_quadform_banded(beatamd_ctx *ctx, const double *wb, int64_t band, int64_t M, int64_t nd, int64_t C, const double *X,
                           int64_t xs_c, int64_t xs_d, double *quad, int64_t q_stride, const int *guard = nullptr, int want = 0);
// several small dense datasets (M <= 512 each) of one residual matrix in one launch, MVN epilogue
// included: LL[c*ld + d] = -0.5 (slog_d + M_d (2 h + log 2pi) + exp(-2h) |W_d x_{c,d}|^2), h = Q[c, hp_off_d]
struct QuadformSmallCall {
    int nd = 0;
    const double *A[8];
    int64_t M[8], xoff[8];
    int upper_tri[8];
    const double *slog[8];
    const int64_t *hp_off[8];
    int64_t C = 0;
    const double *X = nullptr;
    int64_t xs_c = 0;
    const double *Q = nullptr;
    int64_t nparams = 0;
    double *LL = nullptr;
    int64_t ld = 0;
};
bool quadform_small_applicable(int nd, const int64_t *M);
int launch_quadform_small(beatamd_ctx *ctx, const QuadformSmallCall &call);
int launch_check_upper_tri(beatamd_ctx *ctx, const double *A, int64_t nd, int64_t M, int *flag_dev);

// ---- logp.hip (small kernels) ------------------------------------------------------
struct HpSrc {  // hp(c,d) = base[c*stride + (offs ? offs[d] : d)]
    const double *base = nullptr;
    int64_t stride = 0;
    const int64_t *offs = nullptr;
};
// logpts[c*ld + d] = -0.5*(slog[d] + int16(M)*(2hp+log2pi) + (1/exp(2hp))*quad[c*nd+d]*wsq)
int launch_mvn_finish(beatamd_ctx *ctx, int64_t C, int64_t nd, int64_t M, const double *quad,
                      const double *slog, HpSrc hp, double *logpts, int64_t ld);
// scalar-weight quadratic form: quad[c,d] = sum_k (w_d * X[c,d,k])^2
int launch_scalar_quad(beatamd_ctx *ctx, int64_t C, int64_t nd, int64_t M, const double *X,
                       int64_t xs_c, int64_t xs_d, const double *w, double *quad);
// geodetic: mu[c,k] (+)= sum_p slips(c,p) G[p,k]
int launch_geo_stack(beatamd_ctx *ctx, const GeoLib *const *libs, int nvar, int64_t C, const ChainVec *slips, int accumulate,
                     double *mu);
// res[c,k] = (data[k] - mu[c,k]) * odw[k]
int launch_geo_residual(beatamd_ctx *ctx, int64_t C, int64_t Nobs, const double *data,
                        const double *odw, const double *mu, double *res);
// laplacian: out[c*ld] = sum_v -0.5*(-logdet + P*(log2pi+2h) + (1/exp(2h))*quad[c,v])
int launch_laplacian_finish(beatamd_ctx *ctx, int64_t C, int64_t nvar, int64_t P, double logdet,
                            const double *quad, HpSrc hp, double *out, int64_t ld);
// LL[c, nllk-1] = sum of composite sums (problems.py:227-247)
struct LikeGroups {  // composite boundaries inside the llk vector (exclusive ends)
    int32_t end[8];
    int n = 0;
};
// chain_bad (nullable): chains flagged by the index maps / the sweep get like = NaN
int launch_like_sum(beatamd_ctx *ctx, int64_t C, int64_t nllk, const LikeGroups &grp, double *LL,
                    const int32_t *chain_bad);
// likelihood vectors of a target-sharded model from the all-gathered rows of the ranks (k_like_assemble, logp.hip)
int launch_like_assemble(beatamd_ctx *ctx, int64_t C, int64_t nllk, int64_t nsrc, const double *src, const int32_t *dst_col,
                         const double *rest, int64_t rest_ld, int64_t rest_col0, int64_t n_rest, int64_t rest_dst0,
                         const LikeGroups &grp, double *LL, int32_t *chain_bad);
// gather slips of all variables into a dense [C, nvar, P] buffer
int launch_gather_slips(beatamd_ctx *ctx, int64_t C, int nvar, int64_t P, const ChainVec *slips,
                        double *out);
// metropolis.py:276-422 pieces
int launch_propose(beatamd_ctx *ctx, int64_t C, int64_t nparams, const double *Q0,
                   const double *delta, const double *scaling, const double *lower,
                   const double *upper, double *Qprop, int32_t *inbounds);
// grp (nullable): sum the `like` column of Lprop here instead of a launch_like_sum before; acc_sum /
// n_acc (nullable): per-chain and population acceptance counters; advance_step: bump ctx->step_dev
int launch_accept(beatamd_ctx *ctx, int64_t C, int64_t nparams, int64_t nllk, double *Q0,
                  double *L0, const double *Qprop, double *Lprop, const int32_t *inbounds,
                  const double *log_u, double beta, const double *betas, int32_t *accepted,
                  const LikeGroups *grp = nullptr, const int32_t *chain_bad = nullptr,
                  int32_t *acc_sum = nullptr, int64_t *n_acc = nullptr, bool advance_step = false);

// geometry.hip: line-of-sight synthetics of rectangular / Mogi sources, mu [C, Nobs]
// with res (and data, odw [Nobs]): res = (data - mu) * odw is stored instead of mu
int launch_geom_los(beatamd_ctx *ctx, const GeomSources &g, const double *Q, int64_t nparams,
                    int64_t C, double *mu, const double *data = nullptr, const double *odw = nullptr,
                    double *res = nullptr);
// displacement components (n, e, up) per (parameter set, source, point): out [C, nsrc, Nobs, 3]
int launch_geom_disp(beatamd_ctx *ctx, int nsrc, const int32_t *kind, const int64_t *poff,
                     const double *params, int64_t C, int64_t nobs, const double *east,
                     const double *north, double nu, double *out);
// covariance.py:716-771 on device
int launch_autocovariance(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *data,
                          const double *mean, double *out);
int launch_scaled_toeplitz(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *coeffs,
                           const double *stds, double *out);


// ---- gemm.hip: O[m,n] = (sum_k A[m,k] Bop[k,n]) * row_scale[m] on the FP64 matrix cores
struct GemmCall {
    const double *A = nullptr, *B = nullptr, *row_scale = nullptr;
    double *O = nullptr;
    int64_t lda = 0, ldb = 0, ldo = 0, M = 0, N = 0, K = 0;
    int b_kn = 0;     // 0: Bop[k,n] = B[n*ldb + k] ("NT")   1: Bop[k,n] = B[k*ldb + n] ("NN")
    int b_upper = 0;  // NT only: B[n,k] == 0 for k < n
    const char *timer = nullptr;
    // batched / accumulating form (chol.hip): O = alpha * A.Bop (+ O); matrices sA/sB/sO apart
    int nbatch = 1;
    int64_t sA = 0, sB = 0, sO = 0;
    double alpha = 1.0;
    int accumulate = 0;
    int lower_only = 0;   // M == N: tiles strictly above the diagonal are skipped
    int b_lower = 0;      // NN only: Bop[k,n] == 0 for k < n
    int col_block = -1;   // >= 0: only this 128-column block of the product (in-place whitening)
};
int launch_gemm_f64(beatamd_ctx *ctx, const GemmCall &call);
// chol.hip: W = cholesky(inv(C)).T and log det C of a stack of matrices (device pointers)
// notpsd (nullable, device [nbatch]): per-matrix flag instead of the status word for a matrix that is not
// positive definite (its W / log_pdet are then meaningless)
int launch_chol_inverse(beatamd_ctx *ctx, int64_t nbatch, int64_t n, const double *C, double *W, double *log_pdet,
                        int32_t *notpsd = nullptr);
// R [n,n] upper triangular with R^T R = F^T F for a tall F [K,n] (device pointers)
int launch_gram_cholesky(beatamd_ctx *ctx, int64_t K, int64_t n, const double *F, double *R);
// M = Wn . inv(Wo) for stacks of upper-triangular matrices (device pointers)
int launch_triu_ratio(beatamd_ctx *ctx, int64_t nbatch, int64_t n, const double *Wn, const double *Wo, double *M);
// X[b, :] = inv(W[b]) . X[b, :] for upper-triangular W (one vector per matrix, back substitution)
int launch_triu_solve_vec(beatamd_ctx *ctx, int64_t nbatch, int64_t n, const double *W, double *X);

// ---- smc.hip: sampler steps on the device
int launch_smc_calc_beta(beatamd_ctx *ctx, int64_t n, const double *lik, int64_t stride, double beta,
                         double cv, int mode, double dbeta, double *out2, double *weights);
int launch_smc_resample(beatamd_ctx *ctx, int64_t n, const double *weights, double aux, double *cum,
                        int32_t *idx);
int launch_pop_factor(beatamd_ctx *ctx, int64_t n, int64_t np, const double *X, int64_t ldx,
                      const double *w, double *F);
int launch_gather_rows(beatamd_ctx *ctx, int64_t nout, int64_t ncol, const double *src, int64_t lds,
                       int64_t nrow_src, const int32_t *idx, double *out, int64_t ldo);
int launch_tune_scaling(beatamd_ctx *ctx, int64_t C, double *scaling, int32_t *accepted, double interval);
int launch_accumulate_i32(beatamd_ctx *ctx, int64_t C, const int32_t *a, int32_t *acc);
int launch_philox_normal(beatamd_ctx *ctx, double *z, int64_t C, int64_t K, uint64_t seed,
                         uint32_t step, uint64_t first_chain);
int launch_philox_univariate(beatamd_ctx *ctx, double *delta, int64_t C, int64_t np, int kind,
                             const double *scale, uint64_t seed, uint32_t step, uint64_t first_chain);
int launch_step_advance(beatamd_ctx *ctx);
// small parameter vectors (K, np <= 64): draws + factor product + propose in one launch; kind -1
// multivariate (factor [K, np]), 0/1/2 the per-parameter families (factor = scales [np], K == np)
bool draw_propose_applicable(int64_t K, int64_t np);
int launch_draw_propose(beatamd_ctx *ctx, int64_t C, int64_t K, int64_t np, int kind, const double *factor,
                        int df, uint64_t seed, uint32_t step, uint64_t first_chain, const double *Q0,
                        const double *scaling, const double *lower, const double *upper, double *Qprop,
                        double *log_u, int32_t *inbounds);
int launch_philox_chain(beatamd_ctx *ctx, int64_t C, uint64_t seed, uint32_t step, uint64_t first_chain,
                        int df, double *log_u, double *row_scale);

}  // namespace beatamd
