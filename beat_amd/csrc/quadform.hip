// quadform.hip -- chain-batched covariance-weighted misfit  quad[c,d] = || A_d x_{c,d} ||^2
// on the FP64 matrix cores of gfx950 (v_mfma_f64_16x16x4_f64).
//
// Reference arithmetic: beat/models/distributions.py:128-136 (tmp = dot(W_i, r_i);
// dot(tmp, tmp)) with W_i = Covariance.chol_inverse (heart.py:211-237, dense (M,M) with
// upper-triangular content); beat/models/laplacian.py:126-127 (Ls = L.dot(s); Ls.T.dot(Ls)).
//
// One chain is a GEMV (bandwidth bound on W: 134 MB per dataset at M=4096).  Batched over C
// chains it is a GEMM  Y_d = A_d (MxM) * X_d (MxC)  followed by a column-wise squared norm:
// A is read once per 64 chains instead of once per chain, and the arithmetic moves to MFMA.
//
// Tiling: 256-thread workgroup = 4 wavefronts; block tile 64 rows x 128 chains (64 for small
// batches); each wave owns 16 rows x 128 chains = eight 16x16 f64 accumulators (8 VGPRs each).
// K is walked in steps of 16 through a double-buffered LDS stage (row pitch 17 doubles ->
// conflict-free ds_read_b64 of the MFMA operands); the global loads of tile k+1 are issued
// before the MFMAs of tile k, one barrier per tile.  For upper-triangular W the K loop starts
// at the block's first row (half the flops/bytes).  Workgroups are numbered so that the chain
// blocks sharing an A tile run on the same XCD (L2 reuse of W).
// f64 MFMA layouts (cdna_hip_programming.md section 3): A lane l -> A[i=l&15][k=l>>4],
// B lane l -> B[k=l>>4][j=l&15], C/D reg r of lane l -> row (l>>4)+4r, col l&15.
#include "kernels.hpp"

namespace beatamd {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int QF_BM = 64, QF_KB = 16, QF_PITCH = QF_KB + 1;

struct QfArgs {
    const double *A;
    int64_t a_stride, M, nd, C;
    const double *X;
    int64_t xs_c, xs_d;
    int upper_tri;
    double *partial;  // [nd, nrb, C]
    int nrb, ncb;
    int vec_ok;  // rows of A and X are 16-byte aligned -> double2 loads
};

// 4 doubles of row `p` starting at k (zero beyond M)
__device__ __forceinline__ void load4(const double *p, int64_t k, int64_t M, bool row_ok, int vec_ok,
                                      double (&v)[4])
{
    if (row_ok && vec_ok && k + 4 <= M) {
        const double2 a = *reinterpret_cast<const double2 *>(p + k);
        const double2 b = *reinterpret_cast<const double2 *>(p + k + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (row_ok && k + e < M) ? p[k + e] : 0.0;
    }
}

// BC = chains per block (64 or 128); each of the 4 waves owns 16 rows x BC chains
template <int BC>
__global__ void __launch_bounds__(256) k_quadform(QfArgs a)
{
    constexpr int NJ = BC / 16;   // 16x16 accumulators per wave
    constexpr int XR = BC / 64;   // X rows staged per thread
    __shared__ double As[2][QF_BM * QF_PITCH];
    __shared__ double Xs[2][BC * QF_PITCH];
    __shared__ double red[4][BC];

    // XCD-aware decode: workgroups b, b+8, b+16, ... run on the same XCD (round-robin dispatch),
    // so consecutive chain blocks of one (row block, dataset) are placed there: the A tile they
    // share is read from HBM once and served from that XCD's L2 afterwards.
    const int64_t b = blockIdx.x;
    const int64_t nwork = (int64_t)a.nrb * a.nd;          // (row block, dataset) items
    const int64_t q = b / 8, xcd = b % 8;
    const int cb = (int)(q % a.ncb);
    const int64_t item = (q / a.ncb) * 8 + xcd;
    if (item >= nwork) return;
    const int rb = (int)(item % a.nrb), d = (int)(item / a.nrb);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t M = a.M;
    const int64_t i0 = (int64_t)rb * QF_BM, c0 = (int64_t)cb * BC;
    const double *A = a.A + (int64_t)d * a.a_stride;

    const int lr = tid >> 2;        // tile row (A) / chain (X) staged by this thread
    const int lk = (tid & 3) * 4;   // first of its 4 k entries
    const int64_t arow = i0 + lr;
    const bool arow_ok = arow < M;
    const double *Ap = A + arow * M;
    const double *Xp[XR];
    bool x_ok[XR];
#pragma unroll
    for (int r = 0; r < XR; r++) {
        const int64_t ch = c0 + lr + 64 * r;
        x_ok[r] = ch < a.C;
        Xp[r] = a.X + ch * a.xs_c + (int64_t)d * a.xs_d;
    }

    v4f64 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = v4f64{0.0, 0.0, 0.0, 0.0};

    const int64_t kstart = a.upper_tri ? i0 : 0;
    double av[4], xv[XR][4];
    load4(Ap, kstart + lk, M, arow_ok, a.vec_ok, av);
#pragma unroll
    for (int r = 0; r < XR; r++) load4(Xp[r], kstart + lk, M, x_ok[r], a.vec_ok, xv[r]);
    int buf = 0;
    for (int64_t k0 = kstart; k0 < M; k0 += QF_KB) {
        // stage tile k0 (already in registers) into LDS buffer `buf`
#pragma unroll
        for (int e = 0; e < 4; e++) {
            As[buf][lr * QF_PITCH + lk + e] = av[e];
#pragma unroll
            for (int r = 0; r < XR; r++) Xs[buf][(lr + 64 * r) * QF_PITCH + lk + e] = xv[r][e];
        }
        __syncthreads();
        // global loads of the next tile fly during this tile's MFMAs
        if (k0 + QF_KB < M) {
            load4(Ap, k0 + QF_KB + lk, M, arow_ok, a.vec_ok, av);
#pragma unroll
            for (int r = 0; r < XR; r++) load4(Xp[r], k0 + QF_KB + lk, M, x_ok[r], a.vec_ok, xv[r]);
        }
#pragma unroll
        for (int kk = 0; kk < QF_KB / 4; kk++) {
            const int kcol = kk * 4 + (lane >> 4);
            const double aop = As[buf][(wave * 16 + (lane & 15)) * QF_PITCH + kcol];
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const double bop = Xs[buf][(j * 16 + (lane & 15)) * QF_PITCH + kcol];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[j], 0, 0, 0);
            }
        }
        buf ^= 1;  // the other buffer was last read two iterations ago: one barrier per tile
    }
    // column-wise squared norm of this wave's 16 rows
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        double s = acc[j][0] * acc[j][0];
        s = fma(acc[j][1], acc[j][1], s);
        s = fma(acc[j][2], acc[j][2], s);
        s = fma(acc[j][3], acc[j][3], s);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 16) red[wave][j * 16 + lane] = s;
    }
    __syncthreads();
    if (tid < BC) {
        const int64_t c = c0 + tid;
        if (c < a.C) {
            const double s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
            a.partial[((int64_t)d * a.nrb + rb) * a.C + c] = s;
        }
    }
}

__global__ void __launch_bounds__(256) k_quadform_reduce(const double *partial, int64_t C,
                                                        int64_t nd, int nrb, double *quad,
                                                        int64_t q_stride)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C * nd) return;
    const int64_t c = i / nd, d = i - c * nd;
    double s = 0.0;
    for (int rb = 0; rb < nrb; rb++) s += partial[((int64_t)d * nrb + rb) * C + c];
    quad[c * q_stride + d] = s;
}

int launch_quadform(beatamd_ctx *ctx, const QuadformCall &k)
{
    if (k.C == 0 || k.nd == 0) return BEATAMD_OK;
    QfArgs a;
    a.A = k.A; a.a_stride = k.a_stride; a.M = k.M; a.nd = k.nd; a.C = k.C;
    a.X = k.X; a.xs_c = k.xs_c; a.xs_d = k.xs_d;
    a.upper_tri = k.upper_tri;
    a.nrb = (int)((k.M + QF_BM - 1) / QF_BM);
    a.vec_ok = (k.M % 2 == 0) && (k.a_stride % 2 == 0) && (k.xs_c % 2 == 0) && (k.xs_d % 2 == 0) &&
               (((uintptr_t)k.A | (uintptr_t)k.X) % 16 == 0);
    void *p = nullptr;
    BA_TRY(ctx->get_scratch(SL_PARTIAL, (size_t)k.nd * a.nrb * k.C * sizeof(double), &p));
    a.partial = (double *)p;
    const int BC = (k.C > 64) ? 128 : 64;
    a.ncb = (int)((k.C + BC - 1) / BC);
    const int64_t nwork = (int64_t)a.nrb * k.nd;
    const int64_t nblocks = ((nwork + 7) / 8) * 8 * a.ncb;
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "quadform: batch too large");
    {
        ScopedTimer tm(ctx, "quadform");
        if (BC == 128)
            hipLaunchKernelGGL(k_quadform<128>, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(k_quadform<64>, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, a);
    }
    BA_HIP(hipGetLastError());
    const int64_t n = k.C * k.nd;
    hipLaunchKernelGGL(k_quadform_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, a.partial, k.C, k.nd, a.nrb, k.quad, k.q_stride);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// ---- small matrices (geodetic datasets of a few hundred points, geometry mode) ----------------
// All datasets of a composite in ONE launch, the MVN epilogue (distributions.py:119-138) included:
// workgroup = (16 chains, dataset); the residual rows of its chains sit in LDS (pitch = 17 mod 32
// doubles: conflict-free B-operand reads), one wavefront per 16-row tile of W_d (round robin beyond
// 16 tiles), A operands straight from global memory (W_d stays in L2: a few hundred KB), K ascending
// from the tile's first row for upper-triangular W.  Sum of squares per chain: the tiles of a wave in
// ascending order, then the waves in order -- fixed, so a chain's value does not depend on the batch.
constexpr int QS_MAXD = 8, QS_CB = 16;

struct QsArgs {
    int nd;
    const double *A[QS_MAXD];
    int M[QS_MAXD], xoff[QS_MAXD], upper[QS_MAXD];
    const double *slog[QS_MAXD];        // one value each
    const int64_t *hp_off[QS_MAXD];     // one offset into q each
    int64_t C;
    const double *X;    // residuals [C, xs_c]
    int64_t xs_c;
    const double *Q;    // chain states (hyper-parameters)
    int64_t nparams;
    double *LL;         // logpts of dataset d -> LL[c * ld + d]
    int64_t ld;
    int pitch;
};

// QS_KB k-steps (4 columns each) of A operands are in flight while the previous batch multiplies: the
// loads come from L2 with ~1 us latency and a workgroup has few waves, so the prefetch depth, not
// the matrix cores, sets the time
constexpr int QS_MAXW = 16, QS_KBMAX = 16;

template <int QS_KB>
__global__ void __launch_bounds__(64 * QS_MAXW) k_quadform_small(QsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double s_x[];   // [QS_CB][pitch] + red[nwave][QS_CB]
    const int d = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * QS_CB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nthr = blockDim.x, nwave = nthr >> 6;
    const int M = a.M[d], pitch = a.pitch;
    const int Mp = (M + 4 * QS_KB - 1) / (4 * QS_KB) * (4 * QS_KB);   // zero padded to whole batches
    double *red = s_x + QS_CB * pitch;
    for (int i = tid; i < QS_CB * Mp; i += nthr) {
        const int j = i / Mp, k = i - j * Mp;
        const int64_t c = c0 + j;
        s_x[j * pitch + k] = (c < a.C && k < M) ? a.X[c * a.xs_c + a.xoff[d] + k] : 0.0;
    }
    __syncthreads();
    const double *A = a.A[d];
    const int ntile = (M + 15) / 16;
    const int li = lane & 15, lk = lane >> 4;
    double wsum = 0.0;   // chain li
    for (int t = wave; t < ntile; t += nwave) {
        const int r0 = t * 16;
        const int row = r0 + li;
        const bool row_ok = row < M;
        const double *Ap = A + (int64_t)row * M;
        const double *xp = s_x + li * pitch;
        v4f64 acc = v4f64{0.0, 0.0, 0.0, 0.0};
        const int kstart = a.upper[d] ? (r0 / (4 * QS_KB)) * (4 * QS_KB) : 0;
        double av[QS_KB], an[QS_KB];
#pragma unroll
        for (int e = 0; e < QS_KB; e++) {
            const int k = kstart + 4 * e + lk;
            av[e] = (row_ok && k < M) ? Ap[k] : 0.0;
        }
        for (int k0 = kstart; k0 < Mp; k0 += 4 * QS_KB) {
            const int kn = k0 + 4 * QS_KB;
            if (kn < Mp) {
#pragma unroll
                for (int e = 0; e < QS_KB; e++) {
                    const int k = kn + 4 * e + lk;
                    an[e] = (row_ok && k < M) ? Ap[k] : 0.0;
                }
            }
#pragma unroll
            for (int e = 0; e < QS_KB; e++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[e], xp[k0 + 4 * e + lk], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < QS_KB; e++) av[e] = an[e];
        }
        double sq = acc[0] * acc[0];
        sq = fma(acc[1], acc[1], sq);
        sq = fma(acc[2], acc[2], sq);
        sq = fma(acc[3], acc[3], sq);
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        wsum += sq;
    }
    if (lane < 16) red[wave * QS_CB + lane] = wsum;
    __syncthreads();
    if (tid < QS_CB) {
        const int64_t c = c0 + tid;
        if (c < a.C) {
            double quad = 0.0;
            for (int w = 0; w < nwave; w++) quad += red[w * QS_CB + tid];
            const double h = a.Q[c * a.nparams + a.hp_off[d][0]];
            const double norm = (double)(int16_t)M * (2 * h + 1.8378770664093453);
            a.LL[c * a.ld + d] = (-0.5) * (a.slog[d][0] + norm + (1 / exp(h * 2)) * quad);
        }
    }
}

bool quadform_small_applicable(int nd, const int64_t *M)
{
    if (nd < 1 || nd > QS_MAXD) return false;
    for (int d = 0; d < nd; d++)
        if (M[d] < 1 || M[d] > 512) return false;
    return true;
}

int launch_quadform_small(beatamd_ctx *ctx, const QuadformSmallCall &k)
{
    if (k.C == 0 || k.nd == 0) return BEATAMD_OK;
    QsArgs a;
    a.nd = k.nd;
    int64_t mmax = 0;
    for (int d = 0; d < k.nd; d++) {
        a.A[d] = k.A[d]; a.M[d] = (int)k.M[d]; a.xoff[d] = (int)k.xoff[d]; a.upper[d] = k.upper_tri[d];
        a.slog[d] = k.slog[d]; a.hp_off[d] = k.hp_off[d];
        mmax = std::max(mmax, k.M[d]);
    }
    a.C = k.C; a.X = k.X; a.xs_c = k.xs_c; a.Q = k.Q; a.nparams = k.nparams; a.LL = k.LL; a.ld = k.ld;
    static const int kb = getenv("BEATAMD_QS_KB") ? atoi(getenv("BEATAMD_QS_KB")) : 8;
    a.pitch = (int)(((mmax + 4 * QS_KBMAX - 1) / (4 * QS_KBMAX)) * (4 * QS_KBMAX) + 17);   // >= the zero-padded row, = 17 mod 32
    // one wave per 16-row tile of the largest dataset, at most 16 (then tiles round robin)
    const int nwave = (int)std::min<int64_t>(QS_MAXW, (mmax + 15) / 16);
    const size_t lds = ((size_t)QS_CB * a.pitch + (size_t)QS_MAXW * QS_CB) * sizeof(double);
    void (*kern)(QsArgs) = kb == 16 ? k_quadform_small<16> : k_quadform_small<8>;
    if (lds > 64 * 1024)
        BA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ScopedTimer tm(ctx, "quadform");
    hipLaunchKernelGGL(kern, dim3((unsigned)((k.C + QS_CB - 1) / QS_CB), (unsigned)k.nd), dim3(64 * nwave), lds,
                       ctx->stream, a);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// flag_dev[0] is cleared to 0 if any entry strictly below the diagonal is non-zero
__global__ void __launch_bounds__(256) k_check_upper(const double *A, int64_t nd, int64_t M,
                                                    int *flag)
{
    const int64_t total = nd * M * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t r = (i / M) % M, cidx = i % M;
        if (cidx < r && A[i] != 0.0) *flag = 0;
    }
}

int launch_check_upper_tri(beatamd_ctx *ctx, const double *A, int64_t nd, int64_t M, int *flag_dev)
{
    int one = 1;
    BA_HIP(hipMemcpyAsync(flag_dev, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    BA_HIP(hipStreamSynchronize(ctx->stream));  // `one` is a stack temporary
    const int64_t total = nd * M * M;
    unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_check_upper, dim3(grid), dim3(256), 0, ctx->stream, A, nd, M, flag_dev);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// ---- banded upper-triangular operators -----------------------------------------------------------
// The whitening operator W = chol(inv(C)).T of a Markov covariance -- the reference's "exponential" noise structure,
// covariance.py:24-51: C_ij = exp(-|i-j| dt / t0) -- is BIDIAGONAL (inv(C) is tridiagonal); what numpy's inv + cholesky
// leave outside the band is rounding residue, ~2e-15 of the largest entry.  multivariate_normal_chol
// (distributions.py:119-138) then needs two products per sample, not a row of 4096: an HBM-bound pass over the
// residuals instead of an FP64-MFMA GEMM.  A weight set qualifies when every matrix is upper-triangular (exact zeros
// below) and no entry further than `band` <= QF_BAND_LIMIT (16) columns right of the diagonal exceeds 2^-40 of the largest
// entry of ITS ROW (round 6; a matrix-wide maximum before); the dropped terms change a whitened sample by at most M * 2^-40 of its largest term (3.7e-9 at M = 4096
// if they all had one sign; ~6e-11 as rounding residue) -- far inside the path's tolerance (1e-6), and stated in the
// header.  BEATAMD_QF_BAND=0 keeps the dense kernel (A/B and the dense-W bench legs).
constexpr double QF_BAND_EPS = 9.094947017729282e-13;   // 2^-40

// rmax[d*M + r] = bits of max |A_d[r, :]| (non-negative doubles order like their bit patterns; NaN -> a quiet-NaN pattern
// that orders above every number).  One wavefront per row.
__global__ void __launch_bounds__(256) k_band_rowmax(const double *A, int64_t nrows, int64_t M, unsigned long long *rmax)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    unsigned long long m = 0;
    for (int64_t c = threadIdx.x & 63; c < M; c += 64) {
        const double v = fabs(A[row * M + c]);
        const unsigned long long u = (v == v) ? (unsigned long long)__double_as_longlong(v) : 0x7ff8000000000000ull;
        m = u > m ? u : m;
    }
    for (int off = 32; off; off >>= 1) {
        const unsigned long long o = __shfl_down(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) rmax[row] = m;
}

// PASS 0: band[0] = max over all rows of (column - row) of an entry above eps * max|its ROW|  (NaN / inf anywhere: M).
// The threshold is relative to the entry's own row (operators with strongly heterogeneous row scales --
// diag(1 / sigma_i) R -- would lose entries that matter in their row against a matrix-wide maximum).
// PASS 1: dropped[0] = bits of the largest |entry| / max|its row| beyond `band0` columns right of the diagonal (what the banded
// evaluation leaves out; reported by beatamd_weights_band_info).
template <int PASS>
__global__ void __launch_bounds__(256) k_band_width(const double *A, int64_t nrows, int64_t M, const unsigned long long *rmax,
                                                    int *band, int64_t band0, unsigned long long *dropped)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t r = row % M;
    const double big = __longlong_as_double((long long)rmax[row]);
    int b = 0;
    unsigned long long dr = 0;
    if (PASS == 0 && !(big <= 1.79e308)) b = (int)min(M, (int64_t)0x7fffffff);
    const double thr = big * QF_BAND_EPS;
    for (int64_t c = r + 1 + (threadIdx.x & 63); c < M; c += 64) {
        const double v = fabs(A[row * M + c]);
        if (PASS == 0) {
            if (v > thr) b = max(b, (int)(c - r));
        } else if (c - r > band0 && big > 0.0) {
            const unsigned long long u = (unsigned long long)__double_as_longlong(v / big);
            dr = u > dr ? u : dr;
        }
    }
    if (PASS == 0) {
        for (int off = 32; off; off >>= 1) b = max(b, __shfl_down(b, off));
        if ((threadIdx.x & 63) == 0 && b) atomicMax(band, b);
    } else {
        for (int off = 32; off; off >>= 1) {
            const unsigned long long o = __shfl_down(dr, off);
            dr = o > dr ? o : dr;
        }
        if ((threadIdx.x & 63) == 0 && dr) atomicMax(dropped, dr);
    }
}

__global__ void __launch_bounds__(256) k_band_pack(const double *A, int64_t nd, int64_t M, int64_t band, double *wb)
{
    const int64_t n = nd * M * (band + 1);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t k = i % (band + 1), r = (i / (band + 1)) % M, d = i / ((band + 1) * M);
    wb[i] = (r + k < M) ? A[(d * M + r) * M + r + k] : 0.0;
}

int launch_band_detect(beatamd_ctx *ctx, const double *A, int64_t nd, int64_t M, void *scratch, int64_t *band_host,
                       double *dropped_rel_host)
{
    // scratch: [nd*M] uint64 row maxima + one uint64 + one int
    unsigned long long *rmax = (unsigned long long *)scratch;
    unsigned long long *dropped = rmax + nd * M;
    int *band = reinterpret_cast<int *>(dropped + 1);
    BA_HIP(hipMemsetAsync(dropped, 0, 16, ctx->stream));
    const int64_t nrows = nd * M;
    const unsigned gx = (unsigned)((nrows + 3) / 4);
    hipLaunchKernelGGL(k_band_rowmax, dim3(gx), dim3(256), 0, ctx->stream, A, nrows, M, rmax);
    hipLaunchKernelGGL(k_band_width<0>, dim3(gx), dim3(256), 0, ctx->stream, A, nrows, M, rmax, band, (int64_t)0, dropped);
    BA_HIP(hipGetLastError());
    int b = 0;
    BA_HIP(hipMemcpyAsync(&b, band, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    BA_HIP(hipStreamSynchronize(ctx->stream));
    *band_host = b;
    *dropped_rel_host = 0.0;
    if (b <= QF_BAND_LIMIT) {
        hipLaunchKernelGGL(k_band_width<1>, dim3(gx), dim3(256), 0, ctx->stream, A, nrows, M, rmax, band, (int64_t)b, dropped);
        BA_HIP(hipGetLastError());
        unsigned long long u = 0;
        BA_HIP(hipMemcpyAsync(&u, dropped, sizeof(u), hipMemcpyDeviceToHost, ctx->stream));
        BA_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(dropped_rel_host, &u, sizeof(double));
    }
    return BEATAMD_OK;
}

int launch_band_pack(beatamd_ctx *ctx, const double *A, int64_t nd, int64_t M, int64_t band, double *wb)
{
    const int64_t n = nd * M * (band + 1);
    hipLaunchKernelGGL(k_band_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, A, nd, M, band, wb);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// quad[c, d] = sum_i ( sum_{k <= band} wb[d, i, k] x[c, d, i + k] )^2.  Workgroup = (dataset, QB_NC chains); thread <->
// samples i = tid, tid + 256, ...: its band rows stay in L1 / L2 over the chains, the residual row is read once, coalesced
// (the band's neighbours hit the line just read).  Per chain: every thread sums its samples in ascending order, then a fixed
// tree over the wavefront and over the four wavefronts -- the same bits on every launch and every rank.
constexpr int QB_NC = 8;
struct QbArgs {
    const double *wb;
    int64_t M, nd, C, band;
    const double *X;
    int64_t xs_c, xs_d;
    double *quad;
    int64_t q_stride;
    // (nullable) the launch works only when (*guard != 0) == (want != 0): the stand-in behind k_gfstack_runs in mode 3
    const int *guard;
    int want;
};

// BAND1: the bidiagonal case with the loop over the band unrolled (the same two fused multiply-adds per sample)
template <int BAND1>
__global__ void __launch_bounds__(256) k_quadform_banded(QbArgs a)
{
    __shared__ double red[QB_NC][4];
    const int tid = threadIdx.x;
    const int64_t d = blockIdx.y, c0 = (int64_t)blockIdx.x * QB_NC;
    const int nc = (int)min((int64_t)QB_NC, a.C - c0);
    const double *wb = a.wb + d * a.M * (a.band + 1);
    double q[QB_NC];
#pragma unroll
    for (int j = 0; j < QB_NC; j++) q[j] = 0.0;
    for (int64_t i = tid; i < a.M; i += 256) {
        const double *wr = wb + i * (a.band + 1);
        const int kmax = (int)min(a.band, a.M - 1 - i);
        if (BAND1) {
            const double w0 = wr[0], w1 = wr[1];          // (w1 = 0 in the last row)
            double x0[QB_NC], x1[QB_NC];
#pragma unroll
            for (int j = 0; j < QB_NC; j++) {
                const double *x = a.X + (c0 + min(j, nc - 1)) * a.xs_c + d * a.xs_d + i;
                x0[j] = x[0];
                x1[j] = kmax ? x[1] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < QB_NC; j++) {
                double y = fma(w0, x0[j], 0.0);
                if (kmax) y = fma(w1, x1[j], y);
                if (j < nc) q[j] = fma(y, y, q[j]);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < QB_NC; j++) {
            if (j >= nc) break;
            const double *x = a.X + (c0 + j) * a.xs_c + d * a.xs_d + i;
            double y = 0.0;
            for (int k = 0; k <= kmax; k++) y = fma(wr[k], x[k], y);
            q[j] = fma(y, y, q[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < QB_NC; j++) {
        double v = q[j];
        for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
        if ((tid & 63) == 0) red[j][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < nc) a.quad[(c0 + tid) * a.q_stride + d] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// ---- bidiagonal operators in the CANONICAL summation order (round 6) ----------------------------------------------------
// The stacking kernels evaluate this misfit inside their epilogues (GF_RESID_BAND1, gfshared.hip): a lane there holds the 64
// samples of one tile of one chain, so the order every path follows is
//     quad = 0;  for tile k = 0, 1, ...:   quad += q_k;   if (k is not the last tile) quad = fma(yb_k, yb_k, quad)
//     q_k  = sum over the tile's samples i but its last, ascending (fma(y_i, y_i, q)), + the trace's very last sample
//     y_i  = fma(W[i,i+1], r_{i+1}, fma(W[i,i], r_i, 0));  yb_k = y of the tile's last sample (its neighbour = next tile)
// and a chain's misfit has the same bits whichever kernel stacked it (batch size, rank count, fused or not).  This kernel is
// the path of everything without the epilogue (the runs / small-group / streaming kernels, beatamd_mvn_chol_logp_batch):
// workgroup = (dataset, 16 chains) of 128 threads; the operator's band rows of a chunk of 8 tiles sit in LDS beside the 16
// chains' residual rows of the chunk (pitch 65: thread <-> (chain, tile) reads its 64 samples conflict-free; every thread
// has a tile; 76 KB: two workgroups per CU), one thread per chain adds the tiles up.  The rows come in as one index space
// of 16-byte loads, eight in flight per thread.  (The first version -- chunks of 64 tiles, two chains at a time, each a
// handful of loads and a wait -- took 1.6 ms for 512 chains x 64 traces of 4096 samples, six times the kernel it replaced:
// 0.47 ms now; the order of the sums is what it was.)
constexpr int QB1_NC = 16, QB1_CT = 8, QB1_NT = 128;   // chains per workgroup, tiles per chunk (at most), threads
constexpr int QB1_WP = 130, QB1_XP = 65;               // pitches (doubles) of a tile's band rows / residuals in LDS
// ct tiles per chunk (min(8, tiles of a row)), npc chains per pass (128 / ct, at most all sixteen; thread <-> (chain, tile)
// of the pass)
static size_t qb1_lds(int ct, int npc)
{
    return ((size_t)ct * QB1_WP + (size_t)npc * ((size_t)ct * QB1_XP + 1) + 2 * (size_t)npc * ct + QB1_NC) * sizeof(double)
           + (size_t)npc * ct * sizeof(int);
}

__global__ void __launch_bounds__(QB1_NT) k_quadform_band1(QbArgs a, int ct, int npc)
{
    extern __shared__ __attribute__((aligned(16))) double sm1[];
    const int xstride = ct * QB1_XP + 1;
    double *wl = sm1;                                   // [ct][QB1_WP]: (w0, w1) of sample i of tile k at k * 130 + 2 i
    double *xl = wl + ct * QB1_WP;                      // [npc][ct * 65 + 1]
    double *part = xl + npc * xstride;                  // [npc][ct]
    double *ybv = part + npc * ct;                      // [npc][ct]
    double *sacc = ybv + npc * ct;                      // [QB1_NC]
    int *hasb = reinterpret_cast<int *>(sacc + QB1_NC); // [npc][ct]
    const int tid = threadIdx.x, NT = blockDim.x;
    if (a.guard && (*a.guard != 0) != (a.want != 0)) return;
    const int64_t d = blockIdx.y, c0 = (int64_t)blockIdx.x * QB1_NC;
    const int nc = (int)min((int64_t)QB1_NC, a.C - c0);
    const int64_t M = a.M, CH = (int64_t)ct * 64;
    const double *wb = a.wb + d * M * 2;
    if (tid < QB1_NC) sacc[tid] = 0.0;
    for (int64_t i0 = 0; i0 < M; i0 += CH) {
        const int64_t nch = min(CH, M - i0);            // samples of the chunk
        __syncthreads();                                // (the chunk before is done with wl)
        for (int64_t g = tid; g < 2 * nch; g += NT) wl[(g >> 7) * QB1_WP + (g & 127)] = wb[2 * i0 + g];
        for (int j0 = 0; j0 < nc; j0 += npc) {
            __syncthreads();                            // (the pass before is done with xl / part / ybv)
            {
                // the residuals of the pass's chains (+ the first residual of the next chunk: the neighbour of this chunk's
                // last sample) as ONE index space, eight independent loads per thread in flight (chain after chain, each
                // a handful of loads and a wait, was 16 round trips per chunk: most of the kernel's time)
                const int nchn = min(npc, nc - j0);
                const double *xb = a.X + (c0 + j0) * a.xs_c + d * a.xs_d + i0;
                const bool vec = (nch % 2 == 0) && (a.xs_c % 2 == 0) && (((uintptr_t)xb) % 16 == 0);
                if (vec) {
                    const int per2 = (int)(nch / 2), tot2 = nchn * per2;
                    for (int e0 = 0; e0 < tot2; e0 += NT * 8) {
                        double2 v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int e = e0 + u * NT + tid;
                            const int jj = e / per2, g = 2 * (e - jj * per2);
                            v[u] = e < tot2 ? *reinterpret_cast<const double2 *>(xb + (int64_t)jj * a.xs_c + g) : make_double2(0.0, 0.0);
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int e = e0 + u * NT + tid;
                            const int jj = e / per2, g = 2 * (e - jj * per2);
                            if (e < tot2) {
                                double *xr = xl + jj * xstride + (g >> 6) * QB1_XP + (g & 63);
                                xr[0] = v[u].x; xr[1] = v[u].y;
                            }
                        }
                    }
                    if (i0 + nch < M && tid < nchn) xl[tid * xstride + (int)(nch >> 6) * QB1_XP + (int)(nch & 63)] = xb[(int64_t)tid * a.xs_c + nch];
                } else {
                    const int per = (int)nch + (i0 + nch < M ? 1 : 0);
                    const int tot = nchn * per;
                    for (int e0 = 0; e0 < tot; e0 += NT * 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int e = e0 + u * NT + tid;
                            const int jj = e / per, g = e - jj * per;
                            v[u] = e < tot ? xb[(int64_t)jj * a.xs_c + g] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int e = e0 + u * NT + tid;
                            const int jj = e / per, g = e - jj * per;
                            if (e < tot) xl[jj * xstride + (g >> 6) * QB1_XP + (g & 63)] = v[u];
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < npc * ct) {
                const int jj = tid / ct, k = tid - jj * ct;
                const int64_t n0 = i0 + (int64_t)k * 64;
                const int nvalid = (int)min((int64_t)64, M - n0);
                if (j0 + jj < nc && nvalid > 0) {
                    const double *x = xl + jj * xstride + k * QB1_XP;
                    const double *w = wl + k * QB1_WP;
                    const bool trace_end = n0 + nvalid == M;
                    double q = 0.0, ri = x[0];
                    if (nvalid == 64) {
#pragma unroll 9
                        for (int i = 0; i < 63; i++) {     // (a whole tile: the trip count is known, the LDS reads run ahead)
                            const double rn = x[i + 1];
                            double y = fma(w[2 * i], ri, 0.0);
                            y = fma(w[2 * i + 1], rn, y);
                            q = fma(y, y, q);
                            ri = rn;
                        }
                    } else {
                        for (int i = 0; i + 1 < nvalid; i++) {
                            const double rn = x[i + 1];
                            double y = fma(w[2 * i], ri, 0.0);
                            y = fma(w[2 * i + 1], rn, y);
                            q = fma(y, y, q);
                            ri = rn;
                        }
                    }
                    double yb = 0.0;
                    if (trace_end) {
                        const double y = fma(w[2 * (nvalid - 1)], ri, 0.0);
                        q = fma(y, y, q);
                    } else {
                        yb = fma(w[126], ri, 0.0);
                        yb = fma(w[127], x[QB1_XP], yb);     // first residual of the next tile (last tile: of the next chunk)
                    }
                    part[jj * ct + k] = q;
                    ybv[jj * ct + k] = yb;
                    hasb[jj * ct + k] = trace_end ? 0 : 1;
                }
            }
            __syncthreads();
            if (tid < npc && j0 + tid < nc) {
                const int ntl = (int)((nch + 63) / 64);
                double sq = sacc[j0 + tid];
                for (int k = 0; k < ntl; k++) {
                    sq += part[tid * ct + k];
                    if (hasb[tid * ct + k]) sq = fma(ybv[tid * ct + k], ybv[tid * ct + k], sq);
                }
                sacc[j0 + tid] = sq;
            }
        }
    }
    __syncthreads();
    if (tid < nc) a.quad[(c0 + tid) * a.q_stride + d] = sacc[tid];
}

int launch_quadform_banded(beatamd_ctx *ctx, const double *wb, int64_t band, int64_t M, int64_t nd, int64_t C, const double *X,
                           int64_t xs_c, int64_t xs_d, double *quad, int64_t q_stride, const int *guard, int want)
{
    if (C == 0 || nd == 0) return BEATAMD_OK;
    if (band == 1) {
        QbArgs b;
        b.wb = wb; b.M = M; b.nd = nd; b.C = C; b.band = 1;
        b.X = X; b.xs_c = xs_c; b.xs_d = xs_d; b.quad = quad; b.q_stride = q_stride;
        b.guard = guard; b.want = want;
        BA_CHECK(nd <= 65535, BEATAMD_EINVAL, "quadform_banded: too many datasets");
        ScopedTimer tm(ctx, "quadform");
        const int ct = (int)std::min<int64_t>(QB1_CT, (M + 63) / 64);
        const int npc = std::max(2, std::min(QB1_NC, QB1_NT / ct));
        const size_t lds = qb1_lds(ct, npc);
        BA_HIP(hipFuncSetAttribute((const void *)k_quadform_band1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qb1_lds(QB1_CT, QB1_NC)));
        hipLaunchKernelGGL(k_quadform_band1, dim3((unsigned)((C + QB1_NC - 1) / QB1_NC), (unsigned)nd), dim3(QB1_NT), lds,
                           ctx->stream, b, ct, npc);
        BA_HIP(hipGetLastError());
        return BEATAMD_OK;
    }
    QbArgs a;
    a.wb = wb; a.M = M; a.nd = nd; a.C = C; a.band = band;
    a.X = X; a.xs_c = xs_c; a.xs_d = xs_d; a.quad = quad; a.q_stride = q_stride;
    a.guard = nullptr; a.want = 0;
    BA_CHECK(guard == nullptr, BEATAMD_EINVAL, "quadform_banded: a guarded launch exists for the bidiagonal kernel only");
    BA_CHECK(nd <= 65535, BEATAMD_EINVAL, "quadform_banded: too many datasets");
    {
        ScopedTimer tm(ctx, "quadform");
        const dim3 grid((unsigned)((C + QB_NC - 1) / QB_NC), (unsigned)nd);
        hipLaunchKernelGGL(k_quadform_banded<0>, grid, dim3(256), 0, ctx->stream, a);   // (band 1: k_quadform_band1 above)
    }
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd
