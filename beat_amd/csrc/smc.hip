// smc.hip -- the sampler steps around the forward model, on the device: SMC stage transition
// (tempering step, importance weights, systematic resampling, proposal factor), proposal rows
// from an own counter-based generator, per-chain step-size tuning, row gathers for the
// population / replica exchange.  The population never leaves HBM between stages; only the
// new beta (and acceptance counts) go back to the host.
//
// Reference arithmetic (hvasbath/beat):
//   SMC.calc_beta        beat/sampler/smc.py:133-165   bisection on the coefficient of variation
//   SMC.calc_covariance  beat/sampler/smc.py:167-186   np.cov(population, aweights=weights)
//   SMC.resample         beat/sampler/smc.py:290-324   Kitagawa's deterministic resampling
//   proposal draws       beat/sampler/base.py:35-71, 163-186; metropolis.py:289-292
//   step-size tuning     beat/sampler/metropolis.py:294-306 (pymc's tune table)
//
// All reductions run in a fixed order that does not depend on the launch (one workgroup, fixed
// width), so every rank of a multi-GPU run computes bit-identical stage decisions from the same
// gathered arrays.
#include "kernels.hpp"

namespace beatamd {

constexpr int SMC_TB = 1024;  // one workgroup of 16 wavefronts

// deterministic block-wide sum: butterfly inside each wavefront (every lane ends with the same
// value), then the 16 wavefront sums added in wavefront order by every thread
__device__ __forceinline__ double block_sum(double v, double *sh)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();   // previous users of sh are done
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; w++) s += sh[w];
    return s;
}

__device__ __forceinline__ double block_max(double v, double *sh)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double s = sh[0];
    for (int w = 1; w < nw; w++) s = fmax(s, sh[w]);
    return s;
}

// smc.py:133-165.  mode 0: bisection -> out[0] = new beta, out[1] = old beta, weights of the last
// bisection midpoint (exactly what the reference returns: `temp` of the final loop iteration).
// mode 1: weights for a given exponent step dbeta (final stage, smc.py:526-529).
__global__ void __launch_bounds__(SMC_TB) k_smc_calc_beta(const double *lik, int64_t stride, int64_t n,
                                                         double beta, double cv, int mode,
                                                         double dbeta, double *out, double *weights)
{
    __shared__ double sh[SMC_TB / 64];
    const int tid = threadIdx.x;
    double m = -INFINITY;
    for (int64_t i = tid; i < n; i += SMC_TB) m = fmax(m, lik[i * stride]);
    const double lmax = block_max(m, sh);

    double low = beta, up = 2.0, cur = beta, total = 0.0;
    double step = dbeta;
    bool more = (mode == 1) || (up - low > 1e-6);
    while (more) {
        if (mode == 0) {
            cur = (low + up) / 2.0;
            step = cur - beta;
        }
        double s = 0.0;
        for (int64_t i = tid; i < n; i += SMC_TB) s += exp(step * (lik[i * stride] - lmax));
        total = block_sum(s, sh);
        if (mode == 1) break;
        const double mean = total / (double)n;
        double q = 0.0;
        for (int64_t i = tid; i < n; i += SMC_TB) {
            const double d = exp(step * (lik[i * stride] - lmax)) - mean;
            q += d * d;
        }
        const double var = block_sum(q, sh) / (double)n;
        const double cov_temp = sqrt(var) / mean;     // np.std(temp) / np.mean(temp)
        if (cov_temp > cv) up = cur; else low = cur;
        more = up - low > 1e-6;
    }
    for (int64_t i = tid; i < n; i += SMC_TB)
        weights[i] = exp(step * (lik[i * stride] - lmax)) / total;
    if (tid == 0) {
        out[0] = cur;
        out[1] = beta;
    }
}

int launch_smc_calc_beta(beatamd_ctx *ctx, int64_t n, const double *lik, int64_t stride, double beta,
                         double cv, int mode, double dbeta, double *out2, double *weights)
{
    ScopedTimer tm(ctx, "stage");
    hipLaunchKernelGGL(k_smc_calc_beta, dim3(1), dim3(SMC_TB), 0, ctx->stream, lik, stride, n, beta,
                       cv, mode, dbeta, out2, weights);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// smc.py:290-324.  cum_dist = np.cumsum(weights) is a sequential sum: one lane reproduces it
// term by term (bit-exact; 4096 chains ~ 20 us); the children are then placed by binary search:
// the reference's running index j is the smallest j with u[i] <= cum_dist[j], capped at n-1
// (SURVEY A.15 overrun guard), and np.repeat(parents, N_childs) lists exactly those j in order.
__global__ void __launch_bounds__(SMC_TB) k_smc_resample(const double *weights, int64_t n, double aux,
                                                        double *cum, int32_t *idx)
{
    if (threadIdx.x == 0) {
        double c = 0.0;
        int64_t i = 0;
        for (; i + 8 <= n; i += 8) {
            double w[8];
#pragma unroll
            for (int e = 0; e < 8; e++) w[e] = weights[i + e];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                c += w[e];
                cum[i + e] = c;
            }
        }
        for (; i < n; i++) {
            c += weights[i];
            cum[i] = c;
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += SMC_TB) {
        const double u = ((double)i + aux) / (double)n;
        int64_t lo = 0, hi = n - 1;          // answer in [0, n-1]
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (u > cum[mid]) lo = mid + 1; else hi = mid;
        }
        idx[i] = (int32_t)lo;
    }
}

int launch_smc_resample(beatamd_ctx *ctx, int64_t n, const double *weights, double aux, double *cum,
                        int32_t *idx)
{
    ScopedTimer tm(ctx, "stage");
    hipLaunchKernelGGL(k_smc_resample, dim3(1), dim3(SMC_TB), 0, ctx->stream, weights, n, aux, cum, idx);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// Proposal factor of the weighted population (smc.py:167-186 np.cov(X, aweights=w, bias=False)):
// with v1 = sum w, v2 = sum w^2, mean_j = sum_i w_i x_ij / v1 and
//     F[i, j] = sqrt(w_i / (v1 - v2 / v1)) * (x_ij - mean_j)
// F^T F is that covariance, so rows z . F (z standard normal) are N(0, cov) draws without
// forming or factoring an (often singular) nparams x nparams matrix.  One workgroup per 64
// parameter columns: 64 columns x 4 row groups, fixed-order sums.
__global__ void __launch_bounds__(256) k_pop_factor(const double *X, int64_t ldx, const double *w,
                                                   int64_t n, int64_t np, double *F, int *status)
{
    __shared__ double sh[4][64];
    __shared__ double shw[8];
    const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + col;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = tid; i < n; i += 256) {
        const double wi = w[i];
        s1 += wi;
        s2 += wi * wi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
    }
    if (col == 0) { shw[rg] = s1; shw[4 + rg] = s2; }
    __syncthreads();
    const double v1 = ((shw[0] + shw[1]) + shw[2]) + shw[3];
    const double v2 = ((shw[4] + shw[5]) + shw[6]) + shw[7];
    double m = 0.0;
    if (j < np)
        for (int64_t i = rg; i < n; i += 4) m = fma(w[i], X[i * ldx + j], m);
    sh[rg][col] = m;
    __syncthreads();
    const double mean = (((sh[0][col] + sh[1][col]) + sh[2][col]) + sh[3][col]) / v1;
    const double fact = 1.0 / (v1 - v2 / v1);
    // degenerate weights (one chain carries everything: v1 - v2/v1 = 0) or a non-finite population:
    // the reference's calc_covariance raises (smc.py:181-185); here the status word carries it to the
    // next synchronisation
    bool bad = !(v1 - v2 / v1 > 0.0);
    if (j < np)
        for (int64_t i = rg; i < n; i += 4) {
            const double f = sqrt(w[i] * fact) * (X[i * ldx + j] - mean);
            bad = bad || !isfinite(f);
            F[i * np + j] = f;
        }
    if (__any(bad) && (tid & 63) == 0) atomicOr(status, ST_BAD_COV);
}

int launch_pop_factor(beatamd_ctx *ctx, int64_t n, int64_t np, const double *X, int64_t ldx,
                      const double *w, double *F)
{
    if (n == 0 || np == 0) return BEATAMD_OK;
    ScopedTimer tm(ctx, "stage");
    hipLaunchKernelGGL(k_pop_factor, dim3((unsigned)((np + 63) / 64)), dim3(256), 0, ctx->stream, X,
                       ldx, w, n, np, F, ctx->d_status);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// out[i, :] = src[idx[i], :]   (resampled parents -> chain starts, smc.py:188-240 / base.py:541-571;
// replica exchange permutation, pt.py:573-633)
__global__ void __launch_bounds__(256) k_gather_rows(const double *src, int64_t lds, const int32_t *idx,
                                                    int64_t nrow_src, int64_t ncol, double *out,
                                                    int64_t ldo, int *status)
{
    const int64_t r = blockIdx.x;
    int64_t s = idx[r];
    if (s < 0 || s >= nrow_src) {
        if (threadIdx.x == 0) atomicOr(status, ST_INDEX_OOB);
        s = 0;
    }
    for (int64_t k = threadIdx.x; k < ncol; k += 256) out[r * ldo + k] = src[s * lds + k];
}

int launch_gather_rows(beatamd_ctx *ctx, int64_t nout, int64_t ncol, const double *src, int64_t lds,
                       int64_t nrow_src, const int32_t *idx, double *out, int64_t ldo)
{
    if (nout == 0 || ncol == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)nout), dim3(256), 0, ctx->stream, src, lds, idx,
                       nrow_src, ncol, out, ldo, ctx->d_status);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// metropolis.py:294-306 with pymc's tune table (restated from its documentation):
//   acc < 0.001 x0.1 | < 0.05 x0.5 | < 0.2 x0.9 | > 0.95 x10 | > 0.75 x2 | > 0.5 x1.1
__global__ void __launch_bounds__(256) k_tune_scaling(int64_t C, double *scaling, int32_t *accepted,
                                                     double interval)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double acc = (double)accepted[c] / interval;
    double f = 1.0;
    if (acc < 0.001) f = 0.1;
    else if (acc < 0.05) f = 0.5;
    else if (acc < 0.2) f = 0.9;
    else if (acc > 0.95) f = 10.0;
    else if (acc > 0.75) f = 2.0;
    else if (acc > 0.5) f = 1.1;
    scaling[c] = scaling[c] * f;
    accepted[c] = 0;
}

int launch_tune_scaling(beatamd_ctx *ctx, int64_t C, double *scaling, int32_t *accepted, double interval)
{
    if (C == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_tune_scaling, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, ctx->stream, C,
                       scaling, accepted, interval);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// accepted_total[c] += accepted[c]  (per-stage acceptance bookkeeping stays on the device)
__global__ void __launch_bounds__(256) k_accumulate_i32(int64_t C, const int32_t *a, int32_t *acc)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < C) acc[c] += a[c];
}

int launch_accumulate_i32(beatamd_ctx *ctx, int64_t C, const int32_t *a, int32_t *acc)
{
    if (C == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_accumulate_i32, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, ctx->stream,
                       C, a, acc);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11) -- counter-based, so a draw is a pure function of (seed, step, chain, element): the
// proposal rows of a chain do not depend on how chains are sharded over GPUs.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53-bit uniform in (0, 1): ((hi >> 5) * 2^26 + (lo >> 6) + 0.5) * 2^-53
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo)
{
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

// counter layout: (pair index inside the row, global chain id, step, stream); key = seed
//   stream 0: proposal normals z[c, 2j], z[c, 2j+1] from pair j of chain c (Box-Muller cos / sin)
//   stream 1: chi-square normals of the multivariate-t divisor (base.py:35-71)
//   stream 2: Metropolis uniforms
__global__ void __launch_bounds__(256) k_philox_normal(double *z, int64_t C, int64_t K, uint64_t seed,
                                                      uint32_t step, uint64_t first_chain, const uint32_t *step_dev)
{
    if (step_dev) step = *step_dev;   // device-resident step counter (graph replay): replaces the argument
    const int64_t npair = (K + 1) / 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C * npair) return;
    const int64_t c = i / npair, j = i - c * npair;
    const uint64_t gc = first_chain + (uint64_t)c;
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)gc, step, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    const double rad = sqrt(-2.0 * log(u1));
    const double th = 6.283185307179586476925286766559 * u2;
    z[c * K + 2 * j] = rad * cos(th);
    if (2 * j + 1 < K) z[c * K + 2 * j + 1] = rad * sin(th);
}

// per chain: log of the Metropolis uniform (metrop_select) and, for a multivariate-t proposal with
// df degrees of freedom, the row scale 1 / sqrt(chi2(df) / df)  (base.py:63-71)
__global__ void __launch_bounds__(256) k_philox_chain(int64_t C, uint64_t seed, uint32_t step,
                                                     uint64_t first_chain, int df, double *log_u,
                                                     double *row_scale, const uint32_t *step_dev)
{
    if (step_dev) step = *step_dev;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const uint64_t gc = first_chain + (uint64_t)c;
    uint32_t r[4];
    if (log_u) {
        philox4x32_10(0u, (uint32_t)gc, step, 2u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        log_u[c] = log(u53(r[0], r[1]));
    }
    if (row_scale) {
        double x = 0.0;
        for (int m = 0; m < df; m += 2) {
            philox4x32_10((uint32_t)(m / 2), (uint32_t)gc, step, 1u, (uint32_t)seed,
                          (uint32_t)(seed >> 32), r);
            const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
            const double rad = sqrt(-2.0 * log(u1));
            const double th = 6.283185307179586476925286766559 * u2;
            const double g0 = rad * cos(th), g1 = rad * sin(th);
            x += g0 * g0;
            if (m + 1 < df) x += g1 * g1;
        }
        row_scale[c] = 1.0 / sqrt(x / (double)df);
    }
}

// per-parameter proposal families (reference beat/sampler/base.py:129-160): every component of a row is an
// independent draw times the parameter's scale.  Streams 3 / 4 of the same counter layout; one thread
// per (chain, pair of parameters).
//   kind 0  NormalProposal   normal(scale)                                  (Box-Muller pair)
//   kind 1  CauchyProposal   standard_cauchy() * scale = tan(pi (u - 1/2)) * scale
//   kind 2  LaplaceProposal  (standard_exponential() - standard_exponential()) * scale
//   kind 3  PoissonProposal  poisson(lam = scale) - scale        (base.py:150-155; integer steps around zero mean)
// Poisson variate from ONE uniform by inversion (sequential search from k = 0, pmf recurrence p_k = p_{k-1} lam / k).
// Exact in law while exp(-lam) is a normal double; step widths lam <= 500 only: a wider one (or NaN) raises
// ST_BAD_SCALE -> BEATAMD_EINVAL at the next synchronisation (the host side refuses it beforehand,
// beat_amd/sampler/metropolis.py) and the draw is NaN.  The search stops where the cumulative sum stops growing (a
// uniform above the rounded sum, ~1e-13 of the draws at lam near 500, lands on that far-tail k instead of the search cap).
__device__ __forceinline__ double poisson_from_uniform(double u, double lam, int *status)
{
    if (lam == 0.0) return 0.0;
    if (!(lam > 0.0 && lam <= 500.0)) {
        atomicOr(status, ST_BAD_SCALE);
        return __builtin_nan("");
    }
    double p = exp(-lam), F = p;
    int k = 0;
    while (u > F && k < 4096) {
        k++;
        p *= lam / (double)k;
        const double Fn = F + p;
        if (Fn == F && (double)k > lam) break;
        F = Fn;
    }
    return (double)k;
}

__global__ void __launch_bounds__(256) k_philox_univariate(double *delta, int64_t C, int64_t np, int kind,
                                                          const double *scale, uint64_t seed, uint32_t step,
                                                          uint64_t first_chain, const uint32_t *step_dev, int *status)
{
    if (step_dev) step = *step_dev;
    const int64_t npair = (np + 1) / 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C * npair) return;
    const int64_t c = i / npair, j = i - c * npair;
    const uint64_t gc = first_chain + (uint64_t)c;
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)gc, step, 3u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    double a, b;
    if (kind == 0) {
        const double rad = sqrt(-2.0 * log(u1));
        const double th = 6.283185307179586476925286766559 * u2;
        a = rad * cos(th);
        b = rad * sin(th);
    } else if (kind == 1) {
        a = tan(3.14159265358979323846 * (u1 - 0.5));
        b = tan(3.14159265358979323846 * (u2 - 0.5));
    } else if (kind == 2) {
        uint32_t q[4];
        philox4x32_10((uint32_t)j, (uint32_t)gc, step, 4u, (uint32_t)seed, (uint32_t)(seed >> 32), q);
        a = log(u53(q[0], q[1])) - log(u1);     // E1 - E2 with E = -log u
        b = log(u53(q[2], q[3])) - log(u2);
    } else {
        const double la = scale[2 * j], lb = (2 * j + 1 < np) ? scale[2 * j + 1] : 0.0;
        delta[c * np + 2 * j] = poisson_from_uniform(u1, la, status) - la;
        if (2 * j + 1 < np) delta[c * np + 2 * j + 1] = poisson_from_uniform(u2, lb, status) - lb;
        return;
    }
    delta[c * np + 2 * j] = a * scale[2 * j];
    if (2 * j + 1 < np) delta[c * np + 2 * j + 1] = b * scale[2 * j + 1];
}

int launch_philox_univariate(beatamd_ctx *ctx, double *delta, int64_t C, int64_t np, int kind,
                             const double *scale, uint64_t seed, uint32_t step, uint64_t first_chain)
{
    const int64_t n = C * ((np + 1) / 2);
    if (n == 0) return BEATAMD_OK;
    ScopedTimer tm(ctx, "proposal");
    hipLaunchKernelGGL(k_philox_univariate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       delta, C, np, kind, scale, seed, step, first_chain, ctx->step_dev, ctx->d_status);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

int launch_philox_normal(beatamd_ctx *ctx, double *z, int64_t C, int64_t K, uint64_t seed,
                         uint32_t step, uint64_t first_chain)
{
    const int64_t n = C * ((K + 1) / 2);
    if (n == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_philox_normal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, z,
                       C, K, seed, step, first_chain, ctx->step_dev);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

int launch_philox_chain(beatamd_ctx *ctx, int64_t C, uint64_t seed, uint32_t step, uint64_t first_chain,
                        int df, double *log_u, double *row_scale)
{
    if (C == 0) return BEATAMD_OK;
    hipLaunchKernelGGL(k_philox_chain, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, ctx->stream, C,
                       seed, step, first_chain, df, log_u, row_scale, ctx->step_dev);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// ---- small parameter vectors (geometry mode: ~10 parameters, ~1000 chains): the whole proposal of a
// step in ONE launch -- the draws of k_philox_normal / k_philox_chain / k_philox_univariate (same counters,
// so the same numbers), the factor product of the proposal GEMM (k ascending, one FMA chain per component)
// and k_propose (q = q0 + delta * scaling, prior-box test, parked on q0 outside the box).  A workgroup
// serves DP_CB chains; factor, normals and flags sit in LDS.
constexpr int DP_CB = 16, DP_MAX = 64;

struct DrawProposeArgs {
    int64_t C, K, np;
    int kind;                 // -1 multivariate (factor [K, np], df), else the univariate family
    const double *factor;     // [K, np] or the per-parameter scales [np]
    int df;
    uint64_t seed, first_chain;
    uint32_t step;
    const uint32_t *step_dev;
    int *status;
    const double *Q0, *scaling, *lower, *upper;
    double *Qprop, *log_u;
    int32_t *inbounds;
};

__device__ __forceinline__ void box_muller(const uint32_t (&r)[4], double &a, double &b)
{
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    const double rad = sqrt(-2.0 * log(u1));
    const double th = 6.283185307179586476925286766559 * u2;
    a = rad * cos(th);
    b = rad * sin(th);
}

__global__ void __launch_bounds__(256) k_draw_propose(DrawProposeArgs a)
{
    __shared__ double Fs[DP_MAX * DP_MAX];
    __shared__ double zs[DP_CB][DP_MAX];
    __shared__ double rs[DP_CB];
    __shared__ int ok[DP_CB];
    const uint32_t step = a.step_dev ? *a.step_dev : a.step;
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * DP_CB;
    const int nc = (int)min((int64_t)DP_CB, a.C - c0);
    const int K = (int)a.K, np = (int)a.np;
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
    if (a.kind < 0)
        for (int i = tid; i < K * np; i += 256) Fs[i] = a.factor[i];
    const int npair = (K + 1) / 2;
    for (int i = tid; i < nc * npair; i += 256) {
        const int c = i / npair, j = i - c * npair;
        const uint32_t gc = (uint32_t)(a.first_chain + (uint64_t)(c0 + c));
        uint32_t r[4];
        double x, y;
        if (a.kind < 0) {
            philox4x32_10((uint32_t)j, gc, step, 0u, k0, k1, r);
            box_muller(r, x, y);
        } else {
            philox4x32_10((uint32_t)j, gc, step, 3u, k0, k1, r);
            if (a.kind == 0) {
                box_muller(r, x, y);
            } else if (a.kind == 1) {
                x = tan(3.14159265358979323846 * (u53(r[0], r[1]) - 0.5));
                y = tan(3.14159265358979323846 * (u53(r[2], r[3]) - 0.5));
            } else if (a.kind == 2) {
                uint32_t q[4];
                philox4x32_10((uint32_t)j, gc, step, 4u, k0, k1, q);
                x = log(u53(q[0], q[1])) - log(u53(r[0], r[1]));
                y = log(u53(q[2], q[3])) - log(u53(r[2], r[3]));
            }
            if (a.kind == 3) {
                const double la = a.factor[2 * j], lb = (2 * j + 1 < K) ? a.factor[2 * j + 1] : 0.0;
                x = poisson_from_uniform(u53(r[0], r[1]), la, a.status) - la;
                y = poisson_from_uniform(u53(r[2], r[3]), lb, a.status) - lb;
            } else {
                x *= a.factor[2 * j];
                if (2 * j + 1 < K) y *= a.factor[2 * j + 1];
            }
        }
        zs[c][2 * j] = x;
        if (2 * j + 1 < K) zs[c][2 * j + 1] = y;
    }
    if (tid < nc) {
        const uint32_t gc = (uint32_t)(a.first_chain + (uint64_t)(c0 + tid));
        uint32_t r[4];
        philox4x32_10(0u, gc, step, 2u, k0, k1, r);
        a.log_u[c0 + tid] = log(u53(r[0], r[1]));
        double scale = 1.0;
        if (a.kind < 0 && a.df > 0) {
            double x = 0.0;
            for (int m = 0; m < a.df; m += 2) {
                philox4x32_10((uint32_t)(m / 2), gc, step, 1u, k0, k1, r);
                double g0, g1;
                box_muller(r, g0, g1);
                x += g0 * g0;
                if (m + 1 < a.df) x += g1 * g1;
            }
            scale = 1.0 / sqrt(x / (double)a.df);
        }
        rs[tid] = scale;
        ok[tid] = 1;
    }
    __syncthreads();
    for (int i = tid; i < nc * np; i += 256) {
        const int c = i / np, n = i - c * np;
        double o;
        if (a.kind < 0) {
            double acc = 0.0;
            for (int k = 0; k < K; k++) acc = fma(zs[c][k], Fs[k * np + n], acc);
            o = (a.df > 0) ? acc * rs[c] : acc;
        } else {
            o = zs[c][n];
        }
        const int64_t g = (c0 + c) * a.np + n;
        const double q = a.Q0[g] + o * a.scaling[c0 + c];
        a.Qprop[g] = q;
        if (!(q >= a.lower[n] && q <= a.upper[n])) ok[c] = 0;
    }
    __syncthreads();
    for (int i = tid; i < nc * np; i += 256) {
        const int c = i / np;
        if (!ok[c]) {
            const int64_t g = (c0 + c) * a.np + (i - c * np);
            a.Qprop[g] = a.Q0[g];
        }
    }
    if (tid < nc) a.inbounds[c0 + tid] = ok[tid];
}

bool draw_propose_applicable(int64_t K, int64_t np) { return K >= 1 && K <= DP_MAX && np >= 1 && np <= DP_MAX; }

int launch_draw_propose(beatamd_ctx *ctx, int64_t C, int64_t K, int64_t np, int kind, const double *factor,
                        int df, uint64_t seed, uint32_t step, uint64_t first_chain, const double *Q0,
                        const double *scaling, const double *lower, const double *upper, double *Qprop,
                        double *log_u, int32_t *inbounds)
{
    if (C == 0) return BEATAMD_OK;
    DrawProposeArgs a;
    a.C = C; a.K = K; a.np = np; a.kind = kind; a.factor = factor; a.df = df;
    a.seed = seed; a.first_chain = first_chain; a.step = step; a.step_dev = ctx->step_dev;
    a.status = ctx->d_status;
    a.Q0 = Q0; a.scaling = scaling; a.lower = lower; a.upper = upper;
    a.Qprop = Qprop; a.log_u = log_u; a.inbounds = inbounds;
    ScopedTimer tm(ctx, "proposal");
    hipLaunchKernelGGL(k_draw_propose, dim3((unsigned)((C + DP_CB - 1) / DP_CB)), dim3(256), 0, ctx->stream, a);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

__global__ void k_step_advance(uint32_t *step_dev) { *step_dev += 1u; }

// after the draws of a step: the device-resident counter moves on (captured with the step in a graph)
int launch_step_advance(beatamd_ctx *ctx)
{
    if (!ctx->step_dev) return BEATAMD_OK;
    hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(1), 0, ctx->stream, ctx->step_dev);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd
