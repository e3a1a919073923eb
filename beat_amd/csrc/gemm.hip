// gemm.hip -- FP64 GEMM on the matrix cores of gfx950 (v_mfma_f64_16x16x4_f64), the tile scheme
// of quadform.hip with the product stored instead of reduced:
//
//     O[m, n] = (sum_k A[m, k] * Bop[k, n]) * row_scale[m]
//
//   b_kn = 0  ("NT")  Bop[k, n] = B[n*ldb + k]   library whitening: rows' = rows . W^T
//                     (SeismicWavemap.prewhitened; reference seismic.py:1509-1534 keeps W and
//                     multiplies per step, distributions.py:128)
//   b_kn = 1  ("NN")  Bop[k, n] = B[k*ldb + n]   proposal rows: delta = z . F
//                     (metropolis.py:289-292 proposal_dist(n_steps); base.py:163-186)
//
// 256-thread workgroup = 4 wavefronts, block tile 64 rows x 128 columns, each wave 16 rows x 128
// columns = eight 16x16 accumulators; K walked in steps of 16 through a double-buffered LDS stage
// (pitch 17 doubles: conflict-free ds_read_b64 operand fetches), global loads of tile k+1 in
// flight during the MFMAs of tile k, one barrier per tile.  b_upper (NT only): B[n, k] == 0 for
// k < n, so a column block starts its K loop at its first column (chol_inverse is upper
// triangular, heart.py:233) -- half the flops of the whitening.
// f64 MFMA layout: A lane l -> A[i=l&15][k=l>>4], B lane l -> B[k=l>>4][j=l&15],
// C/D reg r of lane l -> row (l>>4)+4r, col l&15.
#include "kernels.hpp"

namespace beatamd {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int GM_BM = 64, GM_BN = 128, GM_KB = 16, GM_PITCH = GM_KB + 1;

struct GemmArgs {
    const double *A, *B, *row_scale;
    double *O;
    int64_t lda, ldb, ldo, M, N, K;
    int b_kn, b_upper, vec_ok;
    int ncb;
    // batched / accumulating form used by the blocked factorisations (chol.hip)
    int64_t sA, sB, sO;   // element strides between the matrices of a batch (blockIdx.y)
    double alpha;         // O = alpha * (A . Bop) [+ O when accumulate]
    int accumulate;
    int lower_only;       // M x N output with M == N: tiles strictly above the diagonal are skipped
    int b_lower;          // NN only: Bop[k, n] == 0 for k < n (lower-triangular B)
    int cb0;              // first column block of this launch (ncb counts the blocks of the launch)
};

__device__ __forceinline__ void gm_load4(const double *p, int64_t k, int64_t K, bool ok, int vec_ok,
                                         double (&v)[4])
{
    if (ok && vec_ok && k + 4 <= K) {
        const double2 a = *reinterpret_cast<const double2 *>(p + k);
        const double2 b = *reinterpret_cast<const double2 *>(p + k + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (ok && k + e < K) ? p[k + e] : 0.0;
    }
}

template <int B_KN>
__global__ void __launch_bounds__(256) k_gemm_f64(GemmArgs a)
{
    constexpr int NJ = GM_BN / 16;
    __shared__ double As[2][GM_BM * GM_PITCH];
    __shared__ double Bs[2][GM_BN * GM_PITCH];

    const int cb = a.cb0 + (int)(blockIdx.x % a.ncb);
    const int64_t rb = blockIdx.x / a.ncb;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t i0 = rb * GM_BM, n0 = (int64_t)cb * GM_BN;
    const int64_t K = a.K;
    if (a.lower_only && n0 > i0 + GM_BM - 1) return;
    a.A += (int64_t)blockIdx.y * a.sA;
    a.B += (int64_t)blockIdx.y * a.sB;
    a.O += (int64_t)blockIdx.y * a.sO;

    // A tile [64 x 16]: thread -> row lr, 4 consecutive k
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const bool a_ok = (i0 + lr) < a.M;
    const double *Ap = a.A + (i0 + lr) * a.lda;
    // B tile NT [128 x 16]: two columns per thread (lr, lr + 64), 4 consecutive k each
    // B tile NN [16 x 128]: thread -> k row bk, 8 consecutive columns from bn
    const int bk = tid >> 4, bn = (tid & 15) * 8;
    const double *Bp0 = nullptr, *Bp1 = nullptr;
    bool b_ok0 = false, b_ok1 = false;
    if (!B_KN) {
        b_ok0 = (n0 + lr) < a.N;
        b_ok1 = (n0 + lr + 64) < a.N;
        Bp0 = a.B + (n0 + lr) * a.ldb;
        Bp1 = a.B + (n0 + lr + 64) * a.ldb;
    }

    v4f64 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = v4f64{0.0, 0.0, 0.0, 0.0};

    int64_t kstart = 0;
    if (!B_KN && a.b_upper) kstart = (n0 / GM_KB) * GM_KB;
    if (B_KN && a.b_lower) kstart = min((n0 / GM_KB) * GM_KB, ((K - 1) / GM_KB) * GM_KB);
    double av[4], bv[8];
    auto load_tile = [&](int64_t k0) {
        gm_load4(Ap, k0 + lk, K, a_ok, a.vec_ok, av);
        if (!B_KN) {
            double t0[4], t1[4];
            gm_load4(Bp0, k0 + lk, K, b_ok0, a.vec_ok, t0);
            gm_load4(Bp1, k0 + lk, K, b_ok1, a.vec_ok, t1);
#pragma unroll
            for (int e = 0; e < 4; e++) { bv[e] = t0[e]; bv[4 + e] = t1[e]; }
        } else {
            const int64_t k = k0 + bk;
            const double *bp = a.B + k * a.ldb + n0 + bn;
#pragma unroll
            for (int e = 0; e < 8; e++) bv[e] = (k < K && n0 + bn + e < a.N) ? bp[e] : 0.0;
        }
    };
    load_tile(kstart);
    int buf = 0;
    for (int64_t k0 = kstart; k0 < K; k0 += GM_KB) {
#pragma unroll
        for (int e = 0; e < 4; e++) As[buf][lr * GM_PITCH + lk + e] = av[e];
        if (!B_KN) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                Bs[buf][lr * GM_PITCH + lk + e] = bv[e];
                Bs[buf][(lr + 64) * GM_PITCH + lk + e] = bv[4 + e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) Bs[buf][(bn + e) * GM_PITCH + bk] = bv[e];
        }
        __syncthreads();
        if (k0 + GM_KB < K) load_tile(k0 + GM_KB);
#pragma unroll
        for (int kk = 0; kk < GM_KB / 4; kk++) {
            const int kcol = kk * 4 + (lane >> 4);
            const double aop = As[buf][(wave * 16 + (lane & 15)) * GM_PITCH + kcol];
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const double bop = Bs[buf][(j * 16 + (lane & 15)) * GM_PITCH + kcol];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[j], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    // store: reg r of lane l -> row (l>>4) + 4r, column l&15 of the wave's 16 x 16 block j
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int64_t row = i0 + wave * 16 + (lane >> 4) + 4 * r;
        if (row >= a.M) continue;
        const double sc = a.row_scale ? a.row_scale[row] : 1.0;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int64_t col = n0 + j * 16 + (lane & 15);
            if (col >= a.N) continue;
            double o = a.row_scale ? acc[j][r] * sc : acc[j][r];
            if (a.alpha != 1.0) o *= a.alpha;   // (alpha = +-1 in every caller: exact)
            if (a.accumulate) o += a.O[row * a.ldo + col];
            a.O[row * a.ldo + col] = o;
        }
    }
}

int launch_gemm_f64(beatamd_ctx *ctx, const GemmCall &k)
{
    if (k.M == 0 || k.N == 0) return BEATAMD_OK;
    BA_CHECK(k.A && k.B && k.O && k.K > 0, BEATAMD_EINVAL, "gemm: bad argument");
    GemmArgs a;
    a.A = k.A; a.B = k.B; a.O = k.O; a.row_scale = k.row_scale;
    a.lda = k.lda; a.ldb = k.ldb; a.ldo = k.ldo;
    a.M = k.M; a.N = k.N; a.K = k.K;
    a.b_kn = k.b_kn;
    a.b_upper = k.b_kn ? 0 : k.b_upper;
    a.vec_ok = (k.lda % 2 == 0) && (k.ldb % 2 == 0) && (((uintptr_t)k.A | (uintptr_t)k.B) % 16 == 0);
    a.sA = k.sA; a.sB = k.sB; a.sO = k.sO;
    a.alpha = k.alpha; a.accumulate = k.accumulate;
    a.lower_only = k.lower_only;
    a.b_lower = k.b_kn ? k.b_lower : 0;
    if (k.nbatch > 1)
        a.vec_ok = a.vec_ok && (k.sA % 2 == 0) && (k.sB % 2 == 0);
    a.ncb = (int)((k.N + GM_BN - 1) / GM_BN);
    a.cb0 = 0;
    if (k.col_block >= 0) {
        // one column block per launch: with an upper-triangular NT operand the tile (rb, cb) reads A[:, k >= n0]
        // only, so launches in ascending block order may write O over A (in-place whitening, capi.cpp)
        BA_CHECK(k.col_block < a.ncb, BEATAMD_EINVAL, "gemm: column block %d of %d", k.col_block, a.ncb);
        a.cb0 = k.col_block;
        a.ncb = 1;
    }
    const int64_t nrb = (k.M + GM_BM - 1) / GM_BM;
    const int64_t nblocks = nrb * a.ncb;
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gemm: too many tiles");
    BA_CHECK(k.nbatch >= 1 && k.nbatch <= 65535, BEATAMD_EINVAL, "gemm: batch of %d", k.nbatch);
    ScopedTimer tm(ctx, k.timer ? k.timer : "gemm");
    const dim3 grid((unsigned)nblocks, (unsigned)k.nbatch);
    if (k.b_kn)
        hipLaunchKernelGGL(k_gemm_f64<1>, grid, dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_gemm_f64<0>, grid, dim3(256), 0, ctx->stream, a);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd
