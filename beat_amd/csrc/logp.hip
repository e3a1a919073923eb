// logp.hip -- the small kernels around the two heavy ones: likelihood epilogues, the
// geodetic static stacking, the Metropolis propose/accept step.  All O(C * small).
#include "kernels.hpp"

namespace beatamd {

#define LOG_2PI 1.8378770664093453  // log(2*pi), distributions.py:13

__device__ __forceinline__ double hp_value(const HpSrc &h, int64_t c, int64_t d)
{
    return h.base[c * h.stride + (h.offs ? h.offs[d] : d)];
}

// distributions.py:119-138:
//   -0.5 * (slog_pdet + M*(2*hp + log_2pi) + (1/exp(hp*2)) * dot(tmp,tmp)),  M cast to int16
__global__ void __launch_bounds__(256) k_mvn_finish(int64_t C, int64_t nd, int64_t M,
                                                   const double *quad, const double *slog,
                                                   HpSrc hp, double *logpts, int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C * nd) return;
    const int64_t c = i / nd, d = i - c * nd;
    const double h = hp_value(hp, c, d);
    const double norm = (double)(int16_t)M * (2 * h + LOG_2PI);
    logpts[c * ld + d] = (-0.5) * (slog[d] + norm + (1 / exp(h * 2)) * quad[i]);
}

int launch_mvn_finish(beatamd_ctx *ctx, int64_t C, int64_t nd, int64_t M, const double *quad,
                      const double *slog, HpSrc hp, double *logpts, int64_t ld)
{
    const int64_t n = C * nd;
    if (n == 0) return BEATAMD_OK;
    ScopedTimer tm(ctx, "finish");
    hipLaunchKernelGGL(k_mvn_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       C, nd, M, quad, slog, hp, logpts, ld);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// one wavefront per (c,d): quad = sum_k (w_d x_k)^2
__global__ void __launch_bounds__(256) k_scalar_quad(int64_t C, int64_t nd, int64_t M,
                                                    const double *X, int64_t xs_c, int64_t xs_d,
                                                    const double *w, double *quad)
{
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= C * nd) return;
    const int64_t c = i / nd, d = i - c * nd;
    const double *x = X + c * xs_c + d * xs_d;
    const double wd = w[d];
    double q = 0.0;
    for (int64_t k = lane; k < M; k += 64) {
        const double t = wd * x[k];
        q = fma(t, t, q);
    }
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
    if (lane == 0) quad[i] = q;
}

int launch_scalar_quad(beatamd_ctx *ctx, int64_t C, int64_t nd, int64_t M, const double *X,
                       int64_t xs_c, int64_t xs_d, const double *w, double *quad)
{
    const int64_t n = C * nd;
    if (n == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_scalar_quad, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, C,
                       nd, M, X, xs_c, xs_d, w, quad);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// ffi/base.py:292-305  mu = G.T.dot(slips), summed over the slip variables (geodetic.py:1065-1070): block (CH chains,
// 128 observation points), the chains' slips in LDS as [variable][p][CH]; an element of G is read once per CH chains and
// sixteen loads are in flight per lane (round 6: one chain per block read the 1.4 MB of G 512 times per call, one launch
// per slip variable with mu through memory in between: 2 x 58 us at 512 chains x 428 points x 400 patches).  Per (chain,
// point) the same fma sequence -- variables ascending, patches ascending -- as the one-chain kernel: bitwise equal.
struct GeoStackArgs {
    const double *G[4];
    ChainVec slips[4];
    int nvar, accumulate;
    int64_t P, Nobs, C;
    double *mu;
};

template <int CH>
__global__ void __launch_bounds__(256) k_geo_stack(GeoStackArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double s_slip[];
    const int64_t c0 = (int64_t)blockIdx.y * CH;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t P = a.P, Nobs = a.Nobs;
    for (int v = 0; v < a.nvar; v++)
        for (int64_t i = threadIdx.x; i < P * CH; i += blockDim.x) {
            const int64_t p = i / CH, j = i % CH;
            s_slip[v * P * CH + i] = c0 + j < a.C ? a.slips[v].base[(c0 + j) * a.slips[v].stride + a.slips[v].off + p] : 0.0;
        }
    __syncthreads();
    if (k >= Nobs) return;
    double acc[CH];
#pragma unroll
    for (int j = 0; j < CH; j++) acc[j] = (a.accumulate && c0 + j < a.C) ? a.mu[(c0 + j) * Nobs + k] : 0.0;
    for (int v = 0; v < a.nvar; v++) {
        const double *G = a.G[v] + k;
        const double *sl = s_slip + v * P * CH;
        // two groups of sixteen loads of G in flight per lane: the next group is requested before the group in hand is
        // used (a block is two wavefronts; the loop is a row of round trips to L2, nothing else covers them)
        int64_t p = 0;
        double ga[16], gb[16];
        const int64_t nfull = P / 16;
        if (nfull > 0) {
#pragma unroll
            for (int u = 0; u < 16; u++) ga[u] = G[u * Nobs];
        }
        for (int64_t ch = 0; ch < nfull; ch += 2) {
            if (ch + 1 < nfull) {
#pragma unroll
                for (int u = 0; u < 16; u++) gb[u] = G[(p + 16 + u) * Nobs];
            }
#pragma unroll
            for (int u = 0; u < 16; u++)
#pragma unroll
                for (int j = 0; j < CH; j++) acc[j] = fma(ga[u], sl[(p + u) * CH + j], acc[j]);
            p += 16;
            if (ch + 1 < nfull) {
                if (ch + 2 < nfull) {
#pragma unroll
                    for (int u = 0; u < 16; u++) ga[u] = G[(p + 16 + u) * Nobs];
                }
#pragma unroll
                for (int u = 0; u < 16; u++)
#pragma unroll
                    for (int j = 0; j < CH; j++) acc[j] = fma(gb[u], sl[(p + u) * CH + j], acc[j]);
                p += 16;
            }
        }
        for (; p < P; p++) {
            const double g = G[p * Nobs];
#pragma unroll
            for (int j = 0; j < CH; j++) acc[j] = fma(g, sl[p * CH + j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < CH; j++)
        if (c0 + j < a.C) a.mu[(c0 + j) * Nobs + k] = acc[j];
}

int launch_geo_stack(beatamd_ctx *ctx, const GeoLib *const *libs, int nvar, int64_t C, const ChainVec *slips, int accumulate,
                     double *mu)
{
    if (C == 0 || nvar == 0) return BEATAMD_OK;
    BA_CHECK(nvar >= 1 && nvar <= 4, BEATAMD_EINVAL, "geo_stack: 1..4 slip variables");
    const GeoLib &lib = *libs[0];
    if (nvar > 1 && (size_t)lib.P * nvar * sizeof(double) > 64 * 1024) {
        // (faults of more than 8192 / nvar patches: the variables' slips do not fit the block's LDS together -- one launch per
        // variable, mu through memory; the same fma sequence)
        for (int v = 0; v < nvar; v++) BA_TRY(launch_geo_stack(ctx, libs + v, 1, C, slips + v, (v > 0 || accumulate) ? 1 : 0, mu));
        return BEATAMD_OK;
    }
    GeoStackArgs a;
    for (int v = 0; v < nvar; v++) {
        BA_CHECK(libs[v]->P == lib.P && libs[v]->Nobs == lib.Nobs, BEATAMD_EINVAL, "geo_stack: the libraries of the slip variables differ in shape");
        a.G[v] = libs[v]->g;
        a.slips[v] = slips[v];
    }
    a.nvar = nvar; a.accumulate = accumulate;
    a.P = lib.P; a.Nobs = lib.Nobs; a.C = C;
    a.mu = mu;
    ScopedTimer tm(ctx, "geostack");
    const int CH = (C >= 64 && (size_t)lib.P * nvar * 4 * sizeof(double) <= 64 * 1024) ? 4 : 1;
    const size_t lds = (size_t)lib.P * nvar * CH * sizeof(double);
    BA_CHECK(lds <= 64 * 1024, BEATAMD_EINVAL, "geo_stack: more than 8192 patch slips per chain");
    BA_CHECK((C + CH - 1) / CH <= 65535, BEATAMD_EINVAL, "geo_stack: at most %d chains per call", 65535 * CH);
    const dim3 grid((unsigned)((lib.Nobs + 127) / 128), (unsigned)((C + CH - 1) / CH));
    if (CH == 4) hipLaunchKernelGGL(k_geo_stack<4>, grid, dim3(128), lds, ctx->stream, a);
    else hipLaunchKernelGGL(k_geo_stack<1>, grid, dim3(128), lds, ctx->stream, a);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// geodetic.py:1072-1074: (sdata - mu) * sodws
__global__ void __launch_bounds__(256) k_geo_residual(int64_t C, int64_t Nobs, const double *data,
                                                     const double *odw, const double *mu,
                                                     double *res)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C * Nobs) return;
    const int64_t k = i % Nobs;
    res[i] = (data[k] - mu[i]) * odw[k];
}

int launch_geo_residual(beatamd_ctx *ctx, int64_t C, int64_t Nobs, const double *data,
                        const double *odw, const double *mu, double *res)
{
    const int64_t n = C * Nobs;
    if (n == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_geo_residual, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, C, Nobs, data, odw, mu, res);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// laplacian.py:88-96 _eval_prior summed over slip variables (:126-139)
__global__ void __launch_bounds__(256) k_laplacian_finish(int64_t C, int64_t nvar, int64_t P,
                                                         double logdet, const double *quad,
                                                         HpSrc hp, double *out, int64_t ld)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double h = hp_value(hp, c, 0);
    double s = 0.0;
    for (int64_t v = 0; v < nvar; v++)
        s += (-0.5) * (-logdet + ((double)P * (LOG_2PI + 2 * h)) +
                       (1.0 / exp(h * 2) * quad[c * nvar + v]));
    out[c * ld] = s;
}

int launch_laplacian_finish(beatamd_ctx *ctx, int64_t C, int64_t nvar, int64_t P, double logdet,
                            const double *quad, HpSrc hp, double *out, int64_t ld)
{
    if (C == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_laplacian_finish, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                       ctx->stream, C, nvar, P, logdet, quad, hp, out, ld);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// problems.py:227-247: like = sum over composites of (composite llk vector).sum()
__global__ void __launch_bounds__(256) k_like_sum(int64_t C, int64_t nllk, LikeGroups grp,
                                                 double *LL, const int32_t *chain_bad)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double *l = LL + c * nllk;
    double total = 0.0;
    int k = 0;
    for (int g = 0; g < grp.n; g++) {
        double s = 0.0;
        for (; k < grp.end[g]; k++) s += l[k];
        total += s;
    }
    // a chain whose start times / durations left the library grid (the reference raises
    // IndexError there) carries NaN: metrop_select rejects it (isfinite test in k_accept)
    if (chain_bad && chain_bad[c]) total = __builtin_nan("");
    l[nllk - 1] = total;
}

int launch_like_sum(beatamd_ctx *ctx, int64_t C, int64_t nllk, const LikeGroups &grp, double *LL,
                    const int32_t *chain_bad)
{
    if (C == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_like_sum, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, ctx->stream, C,
                       nllk, grp, LL, chain_bad);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// Likelihood vectors of a model whose seismic targets are SHARDED over ranks (beat_amd/models/sharded.py; SURVEY 8(e):
// "each GPU owns T/R targets for all chains ... one small collective"): the all-gathered block `src` [nsrc, C] holds,
// rank by rank, the rows a rank contributes -- the logpts of its datasets (dst_col[r] = their column in the full vector)
// and one flag row per rank (dst_col[r] = -1: NaN marks a chain whose times left the library grid on one of THAT rank's
// targets).  Thread <-> chain: scatter the rows into LL [C, nllk], copy the replicated columns (geodetic, Laplacian) from
// this rank's local vector, then `like` in k_like_sum's order (per composite, then over composites; problems.py:227-247)
// -- from the same gathered bits on every rank.
__global__ void __launch_bounds__(256) k_like_assemble(int64_t C, int64_t nllk, int64_t nsrc, const double *src,
                                                      const int32_t *dst_col, const double *rest, int64_t rest_ld,
                                                      int64_t rest_col0, int64_t n_rest, int64_t rest_dst0, LikeGroups grp,
                                                      double *LL, int32_t *chain_bad)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double *l = LL + c * nllk;
    bool bad = false;
    for (int64_t r = 0; r < nsrc; r++) {
        const double v = src[r * C + c];
        const int32_t d = dst_col[r];
        if (d >= 0) l[d] = v;
        else bad = bad || (v != v);
    }
    for (int64_t k = 0; k < n_rest; k++) l[rest_dst0 + k] = rest[c * rest_ld + rest_col0 + k];
    double total = 0.0;
    int k = 0;
    for (int g = 0; g < grp.n; g++) {
        double s = 0.0;
        for (; k < grp.end[g]; k++) s += l[k];
        total += s;
    }
    if (bad) total = __builtin_nan("");
    l[nllk - 1] = total;
    if (chain_bad) chain_bad[c] = bad ? 1 : 0;
}

int launch_like_assemble(beatamd_ctx *ctx, int64_t C, int64_t nllk, int64_t nsrc, const double *src, const int32_t *dst_col,
                         const double *rest, int64_t rest_ld, int64_t rest_col0, int64_t n_rest, int64_t rest_dst0,
                         const LikeGroups &grp, double *LL, int32_t *chain_bad)
{
    if (C == 0) return BEATAMD_OK;
    ScopedTimer tm(ctx, "finish");
    hipLaunchKernelGGL(k_like_assemble, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, ctx->stream, C, nllk, nsrc, src,
                       dst_col, rest, rest_ld, rest_col0, n_rest, rest_dst0, grp, LL, chain_bad);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

struct GatherArgs {
    ChainVec slips[4];
};
__global__ void __launch_bounds__(256) k_gather_slips(int64_t C, int nvar, int64_t P, GatherArgs g,
                                                     double *out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C * nvar * P) return;
    const int64_t p = i % P;
    const int64_t cv = i / P;
    const int v = (int)(cv % nvar);
    const int64_t c = cv / nvar;
    out[i] = g.slips[v].base[c * g.slips[v].stride + g.slips[v].off + p];
}

int launch_gather_slips(beatamd_ctx *ctx, int64_t C, int nvar, int64_t P, const ChainVec *slips,
                        double *out)
{
    GatherArgs g;
    for (int v = 0; v < nvar; v++) g.slips[v] = slips[v];
    const int64_t n = C * nvar * P;
    if (n == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_gather_slips, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       C, nvar, P, g, out);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// metropolis.py:313-343: q = q0 + delta * scaling ; prior_logp finite <=> inside the box
__global__ void __launch_bounds__(256) k_propose(int64_t C, int64_t nparams, const double *Q0,
                                                const double *delta, const double *scaling,
                                                const double *lower, const double *upper,
                                                double *Qprop, int32_t *inbounds)
{
    const int64_t c = blockIdx.x;
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    const double sc = scaling[c];
    int ok = 1;
    for (int64_t k = threadIdx.x; k < nparams; k += 256) {
        const double d = delta[c * nparams + k] * sc;
        const double q = Q0[c * nparams + k] + d;
        Qprop[c * nparams + k] = q;
        if (!(q >= lower[k] && q <= upper[k])) ok = 0;
    }
    if (!ok) s_ok = 0;
    __syncthreads();
    // metropolis.py:341-343,383-385: outside the prior box the forward model is NOT evaluated
    // and the chain stays.  The batch evaluates every chain, so park the rejected chain on its
    // current point (keeps durations/start times inside the library grid).
    if (!s_ok)
        for (int64_t k = threadIdx.x; k < nparams; k += 256)
            Qprop[c * nparams + k] = Q0[c * nparams + k];
    if (threadIdx.x == 0) inbounds[c] = s_ok;
}

int launch_propose(beatamd_ctx *ctx, int64_t C, int64_t nparams, const double *Q0,
                   const double *delta, const double *scaling, const double *lower,
                   const double *upper, double *Qprop, int32_t *inbounds)
{
    if (C == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_propose, dim3((unsigned)C), dim3(256), 0, ctx->stream, C, nparams, Q0,
                       delta, scaling, lower, upper, Qprop, inbounds);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// metropolis.py:344-385 + pymc metrop_select: accept iff in bounds, isfinite(mr), log u < mr.
// One workgroup per chain.  Optional tail work of the step, so that a step needs no further launch:
//   grp.n > 0   the `like` column of the proposal is summed here (k_like_sum's order: per composite,
//               then over composites; chain_bad -> NaN) from an LDS copy of the row
//   acc_sum     per-chain acceptance counter (+= flag), n_acc the population total (+= flags)
//   step_dev    the device-resident Philox step counter moves on (read only by the draw kernels, which
//               precede this launch in stream order)
__global__ void __launch_bounds__(256) k_accept(int64_t C, int64_t nparams, int64_t nllk,
                                               double *Q0, double *L0, const double *Qprop,
                                               double *Lprop, const int32_t *inbounds,
                                               const double *log_u, double beta,
                                               const double *betas, int32_t *accepted, LikeGroups grp,
                                               const int32_t *chain_bad, int32_t *acc_sum,
                                               unsigned long long *n_acc, uint32_t *step_dev)
{
    extern __shared__ __attribute__((aligned(16))) double s_l[];
    const int64_t c = blockIdx.x;
    double lp;
    if (grp.n > 0) {
        for (int64_t k = threadIdx.x; k < nllk - 1; k += 256) s_l[k] = Lprop[c * nllk + k];
        __syncthreads();
        if (threadIdx.x == 0) {
            double total = 0.0;
            int k = 0;
            for (int g = 0; g < grp.n; g++) {
                double s = 0.0;
                for (; k < grp.end[g]; k++) s += s_l[k];
                total += s;
            }
            if (chain_bad && chain_bad[c]) total = __builtin_nan("");
            s_l[nllk - 1] = total;
            Lprop[c * nllk + nllk - 1] = total;
        }
        __syncthreads();
        lp = s_l[nllk - 1];
    } else {
        lp = Lprop[c * nllk + nllk - 1];
    }
    const double b = betas ? betas[c] : beta;  // per-replica beta for parallel tempering
    const double mr = b * (lp - L0[c * nllk + nllk - 1]);
    const bool acc = inbounds[c] && isfinite(mr) && (log_u[c] < mr);
    if (acc) {
        for (int64_t k = threadIdx.x; k < nparams; k += 256) Q0[c * nparams + k] = Qprop[c * nparams + k];
        __syncthreads();  // all lanes have read L0[like] before it is overwritten
        if (grp.n > 0) {
            for (int64_t k = threadIdx.x; k < nllk; k += 256) L0[c * nllk + k] = s_l[k];
        } else {
            for (int64_t k = threadIdx.x; k < nllk; k += 256) L0[c * nllk + k] = Lprop[c * nllk + k];
        }
    }
    if (threadIdx.x == 0) {
        accepted[c] = acc ? 1 : 0;
        if (acc_sum && acc) acc_sum[c] += 1;
        if (n_acc && acc) atomicAdd(n_acc, 1ull);
        if (step_dev && c == 0) *step_dev += 1u;
    }
}

int launch_accept(beatamd_ctx *ctx, int64_t C, int64_t nparams, int64_t nllk, double *Q0,
                  double *L0, const double *Qprop, double *Lprop, const int32_t *inbounds,
                  const double *log_u, double beta, const double *betas, int32_t *accepted,
                  const LikeGroups *grp, const int32_t *chain_bad, int32_t *acc_sum, int64_t *n_acc,
                  bool advance_step)
{
    if (C == 0) return BEATAMD_OK;
    LikeGroups g;
    if (grp) g = *grp;
    const size_t lds = grp ? (size_t)nllk * sizeof(double) : 0;
    BA_CHECK(lds <= 48 * 1024, BEATAMD_EINVAL, "accept: likelihood vector of %lld entries", (long long)nllk);
    ScopedTimer tm(ctx, "astep");
    hipLaunchKernelGGL(k_accept, dim3((unsigned)C), dim3(256), lds, ctx->stream, C, nparams, nllk,
                       Q0, L0, Qprop, Lprop, inbounds, log_u, beta, betas, accepted, g, chain_bad, acc_sum,
                       (unsigned long long *)n_acc, advance_step ? ctx->step_dev : nullptr);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// ---------------------------------------------------------------- noise covariance estimation
// covariance.py:716-736 autocovariance: autocov[j] = (1/n) sum_k (d[j+k]-m)(d[k]-m).  One thread
// per lag walks k in the reference's order with separate multiply and add (no contraction), so the
// result is bitwise the reference's O(n^2) Python loop; the trace sits in LDS.
__global__ void __launch_bounds__(256) k_autocovariance(const double *data, int64_t n,
                                                       const double *mean, double *out)
{
    extern __shared__ __attribute__((aligned(16))) double s_d[];
    const int64_t d = blockIdx.y;
    const double *x = data + d * n;
    const double m = mean[d];
    for (int64_t k = threadIdx.x; k < n; k += 256) s_d[k] = x[k] - m;
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double a = 0.0;
    for (int64_t k = 0; k < n - j; k++) a += s_d[j + k] * s_d[k];
    out[d * n + j] = a / (double)n;
}

int launch_autocovariance(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *data,
                          const double *mean, double *out)
{
    if (nd == 0 || n == 0) return BEATAMD_OK;
    BA_CHECK(n * 8 <= 150 * 1024, BEATAMD_EINVAL, "autocovariance: trace longer than 19200 samples");
    BA_CHECK(nd <= 65535, BEATAMD_EINVAL, "autocovariance: too many datasets");
    const size_t lds = (size_t)n * sizeof(double);
    if (lds > 64 * 1024)
        BA_HIP(hipFuncSetAttribute((const void *)k_autocovariance,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_autocovariance, dim3((unsigned)((n + 255) / 256), (unsigned)nd), dim3(256),
                       lds, ctx->stream, data, n, mean, out);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// covariance.py:739-771: C[i,j] = coeffs[|i-j|] * stds[i] * stds[j]
__global__ void __launch_bounds__(256) k_scaled_toeplitz(int64_t nd, int64_t n, const double *coeffs,
                                                        const double *stds, double *out)
{
    const int64_t total = nd * n * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % n, r = (i / n) % n, d = i / (n * n);
        const int64_t l = r > c ? r - c : c - r;
        out[i] = coeffs[d * n + l] * stds[d * n + r] * stds[d * n + c];
    }
}

int launch_scaled_toeplitz(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *coeffs,
                           const double *stds, double *out)
{
    const int64_t total = nd * n * n;
    if (total == 0) return BEATAMD_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(k_scaled_toeplitz, dim3(grid), dim3(256), 0, ctx->stream, nd, n, coeffs, stds,
                       out);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd
