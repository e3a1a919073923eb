// chol.hip -- the whitening operator of a batch of covariance matrices on the device:
//
//     W = cholesky(inv(C)).T   (upper triangular),   log_pdet = log det C
//
// what heart.Covariance.chol_inverse / .log_pdet hand to the likelihood (reference
// beat/heart.py:216-253; per SMC stage for every dataset of a wavemap, seismic.py:1509-1534).
// The reference inverts C and factors the inverse.  With J the exchange matrix,
// chol(J C J) =: M gives  J C J = M M^T  =>  inv(C) = (J M^-T J)(J M^-1 J)  and J M^-T J is LOWER
// triangular with positive diagonal, i.e. it IS the Cholesky factor of inv(C) (uniqueness):
//
//     W = J . inv(M) . J ,   log det C = 2 sum log diag(M)
//
// one factorisation + one triangular inverse (2/3 n^3 flops) instead of factor, invert, multiply,
// factor (4/3 n^3 and more), and no explicit inv(C).  Blocked with 64 x 64 diagonal blocks:
//   k_chol_flip_pad   A = J C J, padded to a multiple of 64 with an identity block
//   k_chol_diag       factor the diagonal block in LDS (unblocked), its inverse beside it
//   panel             A[i,k] <- A[i,k] . inv(L_kk)^T           k_gemm_f64 NT, in place
//   trailing update   A[i,j] -= A[i,k] . A[j,k]^T  (i >= j > k) k_gemm_f64 NT, lower tiles only
//   inverse by row blocks   X[k,k] = inv(L_kk);  X[k,<k] = -inv(L_kk) . (M[k,<k] . X[<k,<k])
//                                                              two k_gemm_f64 NN (b_lower)
//   k_chol_unflip     W[i,j] = X[n-1-i, n-1-j]
// All GEMMs run batched over the matrices of the stack (grid.y), on the FP64 matrix cores.
#include "kernels.hpp"

namespace beatamd {

constexpr int CH_NB = 64;

// A[b][i][j] = C[b][n-1-i][n-1-j] inside n x n, identity in the padding
__global__ void __launch_bounds__(256) k_chol_flip_pad(const double *C, int64_t n, int64_t np, double *A)
{
    const int64_t b = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= np * np) return;
    const int64_t i = idx / np, j = idx % np;
    double v;
    if (i < n && j < n) v = C[(b * n + (n - 1 - i)) * n + (n - 1 - j)];
    else v = (i == j) ? 1.0 : 0.0;
    A[b * np * np + idx] = v;
}

// W[b][i][j] = X[b][n-1-i][n-1-j]
__global__ void __launch_bounds__(256) k_chol_unflip(const double *X, int64_t n, int64_t np, double *W)
{
    const int64_t b = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t i = idx / n, j = idx % n;
    W[b * n * n + idx] = X[(b * np + (n - 1 - i)) * np + (n - 1 - j)];
}

// One workgroup per matrix: factor the 64 x 64 diagonal block kb (lower triangle of A, already
// carrying the updates of the block columns to its left) in LDS, write L back, its inverse to
// Dinv[b][kb] (full 64 x 64, zeros above the diagonal) and into X's diagonal block, and
// sum_j log L_jj to logd[b][kb].  A pivot <= 0 or NaN raises ST_NOT_PSD.
__global__ void __launch_bounds__(256) k_chol_diag(double *A, int64_t np, int kb, double *Dinv, double *X,
                                                   double *logd, int nblk, int *status, int32_t *notpsd)
{
    constexpr int NB = CH_NB, PITCH = NB + 1;
    __shared__ double L[NB * PITCH];
    __shared__ double Li[NB * PITCH];
    __shared__ int bad;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    double *Ab = A + b * np * np + ((int64_t)kb * NB) * np + (int64_t)kb * NB;
    if (tid == 0) bad = 0;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        L[i * PITCH + j] = (j <= i) ? Ab[(int64_t)i * np + j] : 0.0;
    }
    __syncthreads();
    // right-looking, one column per round: pivot, scale the column, rank-1 update of the rest
    for (int j = 0; j < NB; j++) {
        const double d = L[j * PITCH + j];
        if (!(d > 0.0)) {   // also NaN
            if (tid == 0) bad = 1;
        }
        const double r = sqrt(d);
        __syncthreads();
        if (tid == 0) L[j * PITCH + j] = r;
        for (int i = j + 1 + tid; i < NB; i += 256) L[i * PITCH + j] /= r;
        __syncthreads();
        // A[i][l] -= L[i][j] * L[l][j] for j < l <= i
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, l = e % NB;
            if (l > j && l <= i) L[i * PITCH + l] = fma(-L[i * PITCH + j], L[l * PITCH + j], L[i * PITCH + l]);
        }
        __syncthreads();
    }
    // inverse by forward substitution, one column per thread: L . x = e_c
    for (int e = tid; e < NB * PITCH; e += 256) Li[e] = 0.0;
    __syncthreads();
    if (tid < NB) {
        const int c = tid;
        for (int i = c; i < NB; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int q = c; q < i; q++) s = fma(-L[i * PITCH + q], Li[q * PITCH + c], s);
            Li[i * PITCH + c] = s / L[i * PITCH + i];
        }
    }
    __syncthreads();
    double *Db = Dinv + (b * nblk + kb) * NB * NB;
    double *Xb = X + b * np * np + ((int64_t)kb * NB) * np + (int64_t)kb * NB;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        if (j <= i) Ab[(int64_t)i * np + j] = L[i * PITCH + j];
        Db[e] = Li[i * PITCH + j];
        Xb[(int64_t)i * np + j] = Li[i * PITCH + j];
    }
    if (tid == 0) {
        double s = 0.0;
        for (int j = 0; j < NB; j++) s += log(L[j * PITCH + j]);
        logd[b * nblk + kb] = s;
        // (per-matrix flags when the caller repairs the failing matrices itself, else the status word)
        if (bad) {
            if (notpsd) notpsd[b] = 1;
            else atomicOr(status, ST_NOT_PSD);
        }
    }
}

// log_pdet[b] = 2 * sum_k logd[b][k], summed in block order
__global__ void k_chol_logdet(const double *logd, int nblk, int64_t nbatch, double *out)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbatch) return;
    double s = 0.0;
    for (int k = 0; k < nblk; k++) s += logd[b * nblk + k];
    out[b] = 2.0 * s;
}

// blocked right-looking Cholesky of the lower triangles of A [nbatch][np][np] in place (np a multiple
// of 64): diagonal blocks by k_chol_diag (their inverses to Dinv and to X's diagonal blocks, their
// log-determinants to logd), panels and trailing updates as batched GEMMs
// Two-level blocking: 64-wide inner steps (diagonal block in LDS, panel solve) update only the rest
// of their CH_NBO-wide outer panel; the matrix behind the panel gets ONE trailing update per outer
// panel with K = CH_NBO instead of four with K = 64 (a 64 x 128 tile then runs 16 k-steps per load /
// store of its accumulators instead of 4: the K = 64 updates ran at 12 TF, bound by re-reading the
// trailing matrix).
static int ch_nbo()
{
    static const int v = getenv("BEATAMD_CHOL_NBO") ? std::max(64, atoi(getenv("BEATAMD_CHOL_NBO")) / 64 * 64) : 256;
    return v;
}

static int potrf_lower(beatamd_ctx *ctx, int64_t nbatch, int64_t np, double *A, double *Dinv, double *X, double *logd,
                       int32_t *notpsd = nullptr)
{
    const int nblk = (int)(np / CH_NB);
    const int64_t sM = np * np;
    const int64_t CH_NBO = ch_nbo();
    for (int64_t ko = 0; ko < np; ko += CH_NBO) {
        const int64_t ko_end = std::min<int64_t>(ko + CH_NBO, np);
        for (int kb = (int)(ko / CH_NB); kb < (int)(ko_end / CH_NB); kb++) {
            hipLaunchKernelGGL(k_chol_diag, dim3((unsigned)nbatch), dim3(256), 0, ctx->stream, A, np, kb, Dinv, X,
                               logd, nblk, ctx->d_status, notpsd);
            const int64_t r0 = (int64_t)(kb + 1) * CH_NB, below = np - r0;
            if (below == 0) break;
            GemmCall g;
            // panel: A[r0:, kb] <- A[r0:, kb] . inv(L_kk)^T
            g.A = A + r0 * np + (int64_t)kb * CH_NB; g.lda = np; g.sA = sM;
            g.B = Dinv + (int64_t)kb * CH_NB * CH_NB; g.ldb = CH_NB; g.sB = (int64_t)nblk * CH_NB * CH_NB;
            g.O = A + r0 * np + (int64_t)kb * CH_NB; g.ldo = np; g.sO = sM;
            g.M = below; g.N = CH_NB; g.K = CH_NB; g.b_kn = 0; g.nbatch = (int)nbatch;
            g.timer = nullptr;
            BA_TRY(launch_gemm_f64(ctx, g));
            // inside the outer panel: A[r0:, r0:ko_end] -= P . P[:ko_end - r0]^T with P = A[r0:, kb] (lower tiles)
            const int64_t ncol = ko_end - r0;
            if (ncol > 0) {
                GemmCall u;
                u.A = A + r0 * np + (int64_t)kb * CH_NB; u.lda = np; u.sA = sM;
                u.B = u.A; u.ldb = np; u.sB = sM;
                u.O = A + r0 * np + r0; u.ldo = np; u.sO = sM;
                u.M = below; u.N = ncol; u.K = CH_NB; u.b_kn = 0; u.nbatch = (int)nbatch;
                u.alpha = -1.0; u.accumulate = 1; u.lower_only = 1;
                BA_TRY(launch_gemm_f64(ctx, u));
            }
        }
        // behind the panel: A[ko_end:, ko_end:] -= Q . Q^T with Q = A[ko_end:, ko:ko_end], K = panel width
        const int64_t below = np - ko_end;
        if (below > 0) {
            GemmCall u;
            u.A = A + ko_end * np + ko; u.lda = np; u.sA = sM;
            u.B = u.A; u.ldb = np; u.sB = sM;
            u.O = A + ko_end * np + ko_end; u.ldo = np; u.sO = sM;
            u.M = below; u.N = below; u.K = ko_end - ko; u.b_kn = 0; u.nbatch = (int)nbatch;
            u.alpha = -1.0; u.accumulate = 1; u.lower_only = 1;
            BA_TRY(launch_gemm_f64(ctx, u));
        }
    }
    return BEATAMD_OK;
}

int launch_chol_inverse(beatamd_ctx *ctx, int64_t nbatch, int64_t n, const double *C, double *W, double *log_pdet,
                        int32_t *notpsd)
{
    if (nbatch == 0 || n == 0) return BEATAMD_OK;
    BA_CHECK(C && W && log_pdet && nbatch > 0 && n > 0 && nbatch <= 65535, BEATAMD_EINVAL,
             "chol_inverse: bad argument");
    const int64_t np = (n + CH_NB - 1) / CH_NB * CH_NB;
    const int nblk = (int)(np / CH_NB);
    void *p = nullptr;
    BA_TRY(ctx->get_scratch(SL_CHOL_A, (size_t)nbatch * np * np * 8, &p));
    double *A = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_X, (size_t)nbatch * np * np * 8, &p));
    double *X = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_D, (size_t)nbatch * nblk * CH_NB * CH_NB * 8, &p));
    double *Dinv = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_T, (size_t)nbatch * ch_nbo() * np * 8, &p));
    double *T = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_L, (size_t)nbatch * nblk * 8, &p));
    double *logd = (double *)p;
    ScopedTimer tm(ctx, "chol_inverse");
    BA_HIP(hipMemsetAsync(X, 0, (size_t)nbatch * np * np * 8, ctx->stream));
    {
        const dim3 grid((unsigned)((np * np + 255) / 256), (unsigned)nbatch);
        hipLaunchKernelGGL(k_chol_flip_pad, grid, dim3(256), 0, ctx->stream, C, n, np, A);
    }
    const int64_t sM = np * np;
    if (notpsd) BA_HIP(hipMemsetAsync(notpsd, 0, (size_t)nbatch * sizeof(int32_t), ctx->stream));
    BA_TRY(potrf_lower(ctx, nbatch, np, A, Dinv, X, logd, notpsd));
    // X = inv(M), M lower triangular, by row panels of CH_NBO rows (X's 64 x 64 diagonal blocks are in place
    // already): first the panel's own diagonal block of the inverse by 64-row steps, then everything to its
    // left with two panel-high products,   X[R, :po] = X[R, R] . (-M[R, :po] . X[:po, :po])
    const int64_t nbo = ch_nbo();
    for (int64_t po = 0; po < np; po += nbo) {
        const int64_t pe = std::min<int64_t>(po + nbo, np);
        for (int kb = (int)(po / CH_NB) + 1; kb < (int)(pe / CH_NB); kb++) {
            const int64_t kc = (int64_t)kb * CH_NB, w = kc - po;
            GemmCall g;   // T = -M[kb, po:kc] . X[po:kc, po:kc]
            g.A = A + kc * np + po; g.lda = np; g.sA = sM;
            g.B = X + po * np + po; g.ldb = np; g.sB = sM;
            g.O = T; g.ldo = np; g.sO = nbo * np;
            g.M = CH_NB; g.N = w; g.K = w; g.b_kn = 1; g.b_lower = 1; g.alpha = -1.0; g.nbatch = (int)nbatch;
            BA_TRY(launch_gemm_f64(ctx, g));
            GemmCall h;   // X[kb, po:kc] = inv(L_kk) . T
            h.A = Dinv + (int64_t)kb * CH_NB * CH_NB; h.lda = CH_NB; h.sA = (int64_t)nblk * CH_NB * CH_NB;
            h.B = T; h.ldb = np; h.sB = nbo * np;
            h.O = X + kc * np + po; h.ldo = np; h.sO = sM;
            h.M = CH_NB; h.N = w; h.K = CH_NB; h.b_kn = 1; h.nbatch = (int)nbatch;
            BA_TRY(launch_gemm_f64(ctx, h));
        }
        if (po == 0) continue;
        const int64_t rows = pe - po;
        GemmCall g;   // T = -M[R, :po] . X[:po, :po]
        g.A = A + po * np; g.lda = np; g.sA = sM;
        g.B = X; g.ldb = np; g.sB = sM;
        g.O = T; g.ldo = np; g.sO = nbo * np;
        g.M = rows; g.N = po; g.K = po; g.b_kn = 1; g.b_lower = 1; g.alpha = -1.0; g.nbatch = (int)nbatch;
        BA_TRY(launch_gemm_f64(ctx, g));
        GemmCall h;   // X[R, :po] = X[R, R] . T
        h.A = X + po * np + po; h.lda = np; h.sA = sM;
        h.B = T; h.ldb = np; h.sB = nbo * np;
        h.O = X + po * np; h.ldo = np; h.sO = sM;
        h.M = rows; h.N = po; h.K = rows; h.b_kn = 1; h.nbatch = (int)nbatch;
        BA_TRY(launch_gemm_f64(ctx, h));
    }
    {
        const dim3 grid((unsigned)((n * n + 255) / 256), (unsigned)nbatch);
        hipLaunchKernelGGL(k_chol_unflip, grid, dim3(256), 0, ctx->stream, (const double *)X, n, np, W);
        hipLaunchKernelGGL(k_chol_logdet, dim3((unsigned)((nbatch + 63) / 64)), dim3(64), 0, ctx->stream,
                           (const double *)logd, nblk, nbatch, log_pdet);
    }
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd

// ---------------------------------------------------------------------------------------------
// M = Wn . inv(Wo) for stacks of UPPER-triangular matrices (both are whitening operators
// cholesky(inv(C)).T): the operator that takes rows whitened with Wo to rows whitened with Wn,
//     rows . Wn^T = (rows . Wo^T) . M^T ,
// so a pre-whitened library follows a covariance update in place, without a copy of the
// unwhitened library (seismic.py:1509-1534 update_weights).  Right-side
// triangular solve M . Wo = Wn by 64-wide block columns, M upper triangular:
//     M[:, k] = (Wn[:, k] - M[:, <k] . Wo[<k, k]) . inv(Wo[k, k])
// with the diagonal blocks of Wo inverted in LDS (back substitution, one column per thread) and
// the block products on the FP64 matrix cores (batched k_gemm_f64).
namespace beatamd {

// Dinv[b][kb] = inv(U_kk) of the upper-triangular diagonal block kb of U [np x np]; a zero or
// non-finite pivot raises ST_NOT_PSD
__global__ void __launch_bounds__(256) k_triu_diag_inv(const double *U, int64_t np, int nblk, double *Dinv, int *status)
{
    constexpr int NB = CH_NB, PITCH = NB + 1;
    __shared__ double A[NB * PITCH];
    __shared__ double X[NB * PITCH];
    const int64_t b = blockIdx.y;
    const int kb = blockIdx.x, tid = threadIdx.x;
    const double *Ub = U + b * np * np + ((int64_t)kb * NB) * np + (int64_t)kb * NB;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        A[i * PITCH + j] = (j >= i) ? Ub[(int64_t)i * np + j] : 0.0;
        X[i * PITCH + j] = 0.0;
    }
    __syncthreads();
    if (tid < NB) {
        // column c of the inverse: U x = e_c, x_i = 0 for i > c, back substitution upwards
        const int c = tid;
        bool bad = false;
        for (int i = c; i >= 0; i--) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int q = i + 1; q <= c; q++) s = fma(-A[i * PITCH + q], X[q * PITCH + c], s);
            const double d = A[i * PITCH + i];
            if (!(fabs(d) > 0.0) || !(fabs(d) < __builtin_inf())) bad = true;
            X[i * PITCH + c] = s / d;
        }
        if (bad) atomicOr(status, ST_NOT_PSD);
    }
    __syncthreads();
    double *Db = Dinv + (b * nblk + kb) * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) Db[e] = X[(e / NB) * PITCH + (e % NB)];
}

// P[b][i][j] = (i < n && j < n) ? S[b][i][j] : (i == j)  (identity padding to a multiple of 64)
__global__ void __launch_bounds__(256) k_pad_identity(const double *S, int64_t n, int64_t np, double *P)
{
    const int64_t b = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= np * np) return;
    const int64_t i = idx / np, j = idx % np;
    P[b * np * np + idx] = (i < n && j < n) ? S[(b * n + i) * n + j] : ((i == j) ? 1.0 : 0.0);
}

__global__ void __launch_bounds__(256) k_unpad(const double *P, int64_t n, int64_t np, double *S)
{
    const int64_t b = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t i = idx / n, j = idx % n;
    S[b * n * n + idx] = (j >= i) ? P[(b * np + i) * np + j] : 0.0;
}

int launch_triu_ratio(beatamd_ctx *ctx, int64_t nbatch, int64_t n, const double *Wn, const double *Wo, double *M)
{
    if (nbatch == 0 || n == 0) return BEATAMD_OK;
    BA_CHECK(Wn && Wo && M && nbatch > 0 && n > 0 && nbatch <= 65535, BEATAMD_EINVAL, "triu_ratio: bad argument");
    const int64_t np = (n + CH_NB - 1) / CH_NB * CH_NB;
    const int nblk = (int)(np / CH_NB);
    void *p = nullptr;
    BA_TRY(ctx->get_scratch(SL_CHOL_A, (size_t)nbatch * np * np * 8, &p));
    double *A = (double *)p;      // padded Wo
    BA_TRY(ctx->get_scratch(SL_CHOL_X, (size_t)nbatch * np * np * 8, &p));
    double *X = (double *)p;      // padded Wn, overwritten by M block column by block column
    BA_TRY(ctx->get_scratch(SL_CHOL_D, (size_t)nbatch * nblk * CH_NB * CH_NB * 8, &p));
    double *Dinv = (double *)p;
    ScopedTimer tm(ctx, "triu_ratio");
    const dim3 gp((unsigned)((np * np + 255) / 256), (unsigned)nbatch);
    hipLaunchKernelGGL(k_pad_identity, gp, dim3(256), 0, ctx->stream, Wo, n, np, A);
    hipLaunchKernelGGL(k_pad_identity, gp, dim3(256), 0, ctx->stream, Wn, n, np, X);
    hipLaunchKernelGGL(k_triu_diag_inv, dim3((unsigned)nblk, (unsigned)nbatch), dim3(256), 0, ctx->stream,
                       (const double *)A, np, nblk, Dinv, ctx->d_status);
    const int64_t sM = np * np;
    // Two blocking levels (round 4): the update of a block column by everything to its left is done for TWO block
    // columns at a time -- a 128-column product, the full width of the GEMM tile (the 64-column products of the first
    // version left half of every MFMA tile empty) --, then the pair is finished with 64 x 64 steps.
    auto finish_block = [&](int kb) -> int {   // M[:rows, kb] = X[:rows, kb] . inv(Wo[kb, kb])   (in place: one column block)
        const int64_t kc = (int64_t)kb * CH_NB, rows = kc + CH_NB;
        GemmCall h;
        h.A = X + kc; h.lda = np; h.sA = sM;
        h.B = Dinv + (int64_t)kb * CH_NB * CH_NB; h.ldb = CH_NB; h.sB = (int64_t)nblk * CH_NB * CH_NB;
        h.O = X + kc; h.ldo = np; h.sO = sM;
        h.M = rows; h.N = CH_NB; h.K = CH_NB; h.b_kn = 1; h.nbatch = (int)nbatch;
        return launch_gemm_f64(ctx, h);
    };
    for (int pb = 0; pb < nblk; pb += 2) {
        const int64_t pc = (int64_t)pb * CH_NB;
        const int ncol = (pb + 1 < nblk) ? 2 : 1;               // block columns of this panel
        if (pb > 0) {
            GemmCall g;   // X[:rows, panel] -= M[:rows, :pc] . Wo[:pc, panel]   (rows >= pc of M[:, :pc] are zero)
            g.A = X; g.lda = np; g.sA = sM;
            g.B = A + pc; g.ldb = np; g.sB = sM;
            g.O = X + pc; g.ldo = np; g.sO = sM;
            g.M = pc + ncol * CH_NB; g.N = ncol * CH_NB; g.K = pc; g.b_kn = 1; g.alpha = -1.0; g.accumulate = 1;
            g.nbatch = (int)nbatch;
            BA_TRY(launch_gemm_f64(ctx, g));
        }
        BA_TRY(finish_block(pb));
        if (ncol == 2) {
            const int64_t kc2 = pc + CH_NB;
            GemmCall g;   // X[:rows, second] -= M[:rows, first] . Wo[first, second]
            g.A = X + pc; g.lda = np; g.sA = sM;
            g.B = A + pc * np + kc2; g.ldb = np; g.sB = sM;
            g.O = X + kc2; g.ldo = np; g.sO = sM;
            g.M = kc2 + CH_NB; g.N = CH_NB; g.K = CH_NB; g.b_kn = 1; g.alpha = -1.0; g.accumulate = 1;
            g.nbatch = (int)nbatch;
            BA_TRY(launch_gemm_f64(ctx, g));
            BA_TRY(finish_block(pb + 1));
        }
    }
    const dim3 gu((unsigned)((n * n + 255) / 256), (unsigned)nbatch);
    hipLaunchKernelGGL(k_unpad, gu, dim3(256), 0, ctx->stream, (const double *)X, n, np, M);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// ---------------------------------------------------------------------------------------------
// x_b = inv(W_b) . x_b for upper-triangular W_b [n x n] and ONE vector per matrix (back substitution):
// the residual traces of a pre-whitened model are W_old (d - s); the covariance update needs d - s
// (covariance.py, BatchedCovarianceUpdater).  One workgroup of 1024 threads per matrix; 64-row blocks from the
// bottom up: sixteen wavefronts form the dot products of the block's rows with the part of x already solved
// (W rows are contiguous: coalesced, every element of the triangle read once = 67 MB at n = 4096), one wavefront
// then solves the 64 x 64 diagonal block from LDS.  A zero on the diagonal raises BEATAMD_ENOTPSD.
constexpr int TS_B = 64;

__global__ void __launch_bounds__(1024) k_triu_solve_vec(const double *W, int64_t n, double *X, int *status)
{
    extern __shared__ double ts_x[];            // [n] the solution so far, then [TS_B] partial sums, [TS_B*TS_B] block
    double *part = ts_x + n;
    double *blk = part + TS_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *Wb = W + (int64_t)blockIdx.x * n * n;
    double *xb = X + (int64_t)blockIdx.x * n;
    for (int64_t i = tid; i < n; i += 1024) ts_x[i] = xb[i];
    __syncthreads();
    const int64_t nblk = (n + TS_B - 1) / TS_B;
    for (int64_t kb = nblk - 1; kb >= 0; kb--) {
        const int64_t i0 = kb * TS_B, i1 = min(n, i0 + TS_B), j0 = i1;
        // (1) part[r] = sum_{j >= j0} W[i0 + r, j] x[j]; wavefront w takes rows w, w + 16, ...
        for (int r = wave; r < (int)(i1 - i0); r += 16) {
            const double *row = Wb + (i0 + r) * n;
            double acc = 0.0;
            for (int64_t j = j0 + lane; j < n; j += 64) acc = fma(row[j], ts_x[j], acc);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) part[r] = acc;
        }
        // the diagonal block into LDS
        for (int e = tid; e < TS_B * TS_B; e += 1024) {
            const int r = e / TS_B, c = e % TS_B;
            blk[e] = (i0 + r < n && i0 + c < n) ? Wb[(i0 + r) * n + i0 + c] : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        // (2) back substitution inside the block by wavefront 0: lane <-> column
        if (wave == 0) {
            const int m = (int)(i1 - i0);
            double xl = 0.0;                         // x of column `lane` once solved
            for (int r = m - 1; r >= 0; r--) {
                double t = (lane > r && lane < m) ? blk[r * TS_B + lane] * xl : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
                const double d = blk[r * TS_B + r];
                if (!(fabs(d) > 0.0) && lane == 0) atomicOr(status, ST_NOT_PSD);   // (zero or NaN pivot)
                const double xr = (ts_x[i0 + r] - part[r] - t) / d;
                if (lane == r) xl = xr;
            }
            if (lane < m) ts_x[i0 + lane] = xl;
        }
        __syncthreads();
    }
    for (int64_t i = tid; i < n; i += 1024) xb[i] = ts_x[i];
}

int launch_triu_solve_vec(beatamd_ctx *ctx, int64_t nbatch, int64_t n, const double *W, double *X)
{
    if (nbatch == 0 || n == 0) return BEATAMD_OK;
    BA_CHECK(W && X && nbatch > 0 && n > 0, BEATAMD_EINVAL, "triu_solve: bad argument");
    const size_t lds = (size_t)(n + TS_B + TS_B * TS_B) * sizeof(double);
    BA_CHECK(lds <= 160 * 1024, BEATAMD_EINVAL, "triu_solve: n = %lld does not fit the LDS (max %d)", (long long)n,
             (int)(160 * 1024 / 8 - TS_B - TS_B * TS_B));
    ScopedTimer tm(ctx, "triu_solve");
    BA_HIP(hipFuncSetAttribute((const void *)k_triu_solve_vec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_triu_solve_vec, dim3((unsigned)nbatch), dim3(1024), lds, ctx->stream, W, n, X, ctx->d_status);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd

// ---------------------------------------------------------------------------------------------
// R [n x n] upper triangular with R^T R = F^T F for a tall factor F [K x n] (K >> n): the compact
// form of a proposal factor.  SMC draws its stage proposals as z . F with F the weighted, centred
// population (K = number of chains), which needs K normals per proposal row; with many more chains
// than parameters the n x n Cholesky factor of the same covariance gives the same distribution
// from n normals per row (base.py:163-186 factors the covariance the same way).  Gram matrix and
// factorisation on the FP64 matrix cores; a Gram matrix that is not numerically positive definite
// (collapsed population) is BEATAMD_ENOTPSD and the caller keeps the tall factor.
namespace beatamd {

__global__ void __launch_bounds__(256) k_transpose(const double *F, int64_t K, int64_t n, double *Ft)
{
    __shared__ double tile[16][17];
    const int64_t k0 = (int64_t)blockIdx.x * 16, j0 = (int64_t)blockIdx.y * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (k0 + ty < K && j0 + tx < n) tile[ty][tx] = F[(k0 + ty) * n + j0 + tx];
    __syncthreads();
    if (j0 + ty < n && k0 + tx < K) Ft[(j0 + ty) * K + k0 + tx] = tile[tx][ty];
}

// R[i][j] = (j >= i) ? L[j][i] : 0  from the padded lower factor
__global__ void __launch_bounds__(256) k_lower_to_upper(const double *L, int64_t n, int64_t np, double *R)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t i = idx / n, j = idx % n;
    R[idx] = (j >= i) ? L[j * np + i] : 0.0;
}

__global__ void __launch_bounds__(256) k_identity_pad(double *A, int64_t n, int64_t np)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= np * np) return;
    const int64_t i = idx / np, j = idx % np;
    if (i >= n || j >= n) A[idx] = (i == j) ? 1.0 : 0.0;
}

int launch_gram_cholesky(beatamd_ctx *ctx, int64_t K, int64_t n, const double *F, double *R)
{
    BA_CHECK(F && R && K > 0 && n > 0, BEATAMD_EINVAL, "gram_cholesky: bad argument");
    const int64_t np = (n + CH_NB - 1) / CH_NB * CH_NB;
    const int nblk = (int)(np / CH_NB);
    void *p = nullptr;
    BA_TRY(ctx->get_scratch(SL_CHOL_A, (size_t)np * np * 8, &p));
    double *A = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_X, (size_t)np * np * 8, &p));
    double *X = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_D, (size_t)nblk * CH_NB * CH_NB * 8, &p));
    double *Dinv = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_T, (size_t)n * K * 8, &p));
    double *Ft = (double *)p;
    BA_TRY(ctx->get_scratch(SL_CHOL_L, (size_t)nblk * 8, &p));
    double *logd = (double *)p;
    ScopedTimer tm(ctx, "gram_cholesky");
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((K + 15) / 16), (unsigned)((n + 15) / 16)), dim3(256), 0,
                       ctx->stream, F, K, n, Ft);
    GemmCall g;   // A[:n, :n] = Ft . Ft^T
    g.A = Ft; g.lda = K; g.B = Ft; g.ldb = K; g.O = A; g.ldo = np;
    g.M = n; g.N = n; g.K = K; g.b_kn = 0;
    BA_TRY(launch_gemm_f64(ctx, g));
    hipLaunchKernelGGL(k_identity_pad, dim3((unsigned)((np * np + 255) / 256)), dim3(256), 0, ctx->stream, A, n, np);
    BA_TRY(potrf_lower(ctx, 1, np, A, Dinv, X, logd));
    hipLaunchKernelGGL(k_lower_to_upper, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double *)A, n, np, R);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd
