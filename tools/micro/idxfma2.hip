// micro-benchmark 2: CW=16 chains x 4 samples/lane, g rows in a pinned register array selected by M0,
// per-step metadata (16 weights + 16 packed register offsets) prefetched one step ahead by s_load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v16d __attribute__((ext_vector_type(16)));
typedef double d8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// 8 entries: chains e0..e0+7; idx bytes packed in two dwords
#define FMA4(k) \
    "v_fma_f64 %[a" #k "0], %[w" #k "], v[128:129], %[a" #k "0]\n\t" \
    "v_fma_f64 %[a" #k "1], %[w" #k "], v[130:131], %[a" #k "1]\n\t" \
    "v_fma_f64 %[a" #k "2], %[w" #k "], v[132:133], %[a" #k "2]\n\t" \
    "v_fma_f64 %[a" #k "3], %[w" #k "], v[134:135], %[a" #k "3]\n\t"
#define ENT(k, word, sh) "s_bfe_u32 s101, %[" #word "], " #sh "\n\t" "s_set_gpr_idx_idx s101\n\t" FMA4(k)
// pipelined: the index of entry k+1 is extracted (into the other temp) before the FMAs of entry k
#define ENTP(k, tcur, tnext, word, sh) \
    "s_set_gpr_idx_idx " #tcur "\n\t" "s_bfe_u32 " #tnext ", %[" #word "], " #sh "\n\t" FMA4(k)
#define ENTL(k, tcur) "s_set_gpr_idx_idx " #tcur "\n\t" FMA4(k)
#define ACC(k) [a##k##0] "+v"(acc[e0 + k][0]), [a##k##1] "+v"(acc[e0 + k][1]), [a##k##2] "+v"(acc[e0 + k][2]), [a##k##3] "+v"(acc[e0 + k][3])
#define OUTS ACC(0), ACC(1), ACC(2), ACC(3), ACC(4), ACC(5), ACC(6), ACC(7)
#define INS [w0] "s"(w[0]), [w1] "s"(w[1]), [w2] "s"(w[2]), [w3] "s"(w[3]), [w4] "s"(w[4]), [w5] "s"(w[5]), \
          [w6] "s"(w[6]), [w7] "s"(w[7]), [ia] "s"(ia), [ib] "s"(ib), "{v[128:159]}"(R0), "{v[160:191]}"(R1)

template <int VAR>
__device__ __forceinline__ void fma8(double (&acc)[16][4], const int e0, const d8 w, uint32_t ia, uint32_t ib,
                                     const v16d &R0, const v16d &R1)
{
    if (VAR == 0) {
        asm volatile(
            "s_set_gpr_idx_on %[ia], 0x2\n\t"
            ENT(0, ia, 0x80000) ENT(1, ia, 0x80008) ENT(2, ia, 0x80010) ENT(3, ia, 0x80018)
            ENT(4, ib, 0x80000) ENT(5, ib, 0x80008) ENT(6, ib, 0x80010) ENT(7, ib, 0x80018)
            "s_set_gpr_idx_off"
            : OUTS : INS : "s101", "scc");
    } else if (VAR == 1) {
        asm volatile(
            "s_bfe_u32 s100, %[ia], 0x80000\n\t"
            "s_set_gpr_idx_on s100, 0x2\n\t"
            ENTP(0, s100, s101, ia, 0x80008) ENTP(1, s101, s100, ia, 0x80010) ENTP(2, s100, s101, ia, 0x80018)
            ENTP(3, s101, s100, ib, 0x80000) ENTP(4, s100, s101, ib, 0x80008) ENTP(5, s101, s100, ib, 0x80010)
            ENTP(6, s100, s101, ib, 0x80018) ENTL(7, s101)
            "s_set_gpr_idx_off"
            : OUTS : INS : "s100", "s101", "scc");
    } else if (VAR == 2) {   // no index change at all (upper bound of the FMA + s_load pipeline)
        asm volatile(
            "s_set_gpr_idx_on %[ia], 0x2\n\t"
            FMA4(0) FMA4(1) FMA4(2) FMA4(3) FMA4(4) FMA4(5) FMA4(6) FMA4(7)
            "s_set_gpr_idx_off"
            : OUTS : INS : "s101", "scc");
    } else if (VAR == 4 || VAR == 5) {   // no idx mode at all
        asm volatile(
            FMA4(0) FMA4(1) FMA4(2) FMA4(3) FMA4(4) FMA4(5) FMA4(6) FMA4(7)
            : OUTS : INS : "s101", "scc");
    } else {                 // idx straight from the SGPR word (no extraction): wrong slots, timing only
        asm volatile(
            "s_set_gpr_idx_on %[ia], 0x2\n\t"
            ENTL(0, %[ia]) ENTL(1, %[ib]) ENTL(2, %[ia]) ENTL(3, %[ib]) ENTL(4, %[ia]) ENTL(5, %[ib]) ENTL(6, %[ia])
            ENTL(7, %[ib])
            "s_set_gpr_idx_off"
            : OUTS : INS : "s101", "scc");
    }
}

template <int VAR>
__global__ void __launch_bounds__(512) k(const double *__restrict__ rows, const u4 *__restrict__ idx,
                                         const d8 *__restrict__ w, double *__restrict__ out, int P, int nshare)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double acc[16][4];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e][0] = acc[e][1] = acc[e][2] = acc[e][3] = 0.0;
    const int stream = (blockIdx.x / nshare) * 8 + wave;   // nshare blocks (sample tiles) share metadata
    const u4 *ip = idx + (size_t)stream * P;
    const d8 *wp = w + (size_t)stream * P * 2;
    // 8 slots x 4 samples in v[128:191] (2 tuples), loaded once
    v16d R0, R1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double4 a = *reinterpret_cast<const double4 *>(rows + ((size_t)j * 64 + lane) * 4);
        const double4 b = *reinterpret_cast<const double4 *>(rows + ((size_t)(4 + j) * 64 + lane) * 4);
        R0[4 * j] = a.x; R0[4 * j + 1] = a.y; R0[4 * j + 2] = a.z; R0[4 * j + 3] = a.w;
        R1[4 * j] = b.x; R1[4 * j + 1] = b.y; R1[4 * j + 2] = b.z; R1[4 * j + 3] = b.w;
    }
    u4 ic = ip[0];
    d8 wa = wp[0], wb = wp[1];
    for (int p = 0; p < P; p++) {
        // consume point: forces the wait for (ic, wa, wb) here, before the prefetch is issued
        asm volatile("" :: "s"(ic), "s"(wa), "s"(wb));
        const int pn = (VAR == 5) ? 0 : min(p + 1, P - 1);
        const u4 in_ = ip[pn];
        const d8 wan = wp[2 * pn], wbn = wp[2 * pn + 1];
        fma8<VAR>(acc, 0, wa, ic[0], ic[1], R0, R1);
        fma8<VAR>(acc, 8, wb, ic[2], ic[3], R0, R1);
        ic = in_; wa = wan; wb = wbn;
    }
    double *o = out + (size_t)(blockIdx.x * 8 + wave) * 16 * 256;
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
        for (int s = 0; s < 4; s++) o[e * 256 + lane * 4 + s] = acc[e][s];
}

int main(int argc, char **argv)
{
    const int P = 400, NB = 2048, nshare = (argc > 1 ? atoi(argv[1]) : 16), NS = NB / nshare * 8;
    std::vector<double> rows((size_t)8 * 256), w((size_t)NS * P * 16);
    std::vector<uint32_t> idx((size_t)NS * P * 4);
    srand(1);
    for (auto &x : rows) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : w) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : idx) {
        x = 0;
        for (int b = 0; b < 4; b++) x |= (uint32_t)((rand() % 8) * 8) << (8 * b);  // slot * 8 VGPRs
    }
    double *d_rows, *d_w, *d_out;
    uint32_t *d_idx;
    CK(hipMalloc(&d_rows, rows.size() * 8));
    CK(hipMalloc(&d_w, w.size() * 8));
    CK(hipMalloc(&d_idx, idx.size() * 4));
    CK(hipMalloc(&d_out, (size_t)NB * 8 * 16 * 256 * 8));
    CK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int var = 0; var < 6; var++) {
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        auto kk = var == 0 ? k<0> : var == 1 ? k<1> : var == 2 ? k<2> : var == 3 ? k<3> : var == 4 ? k<4> : k<5>;
        hipLaunchKernelGGL(kk, dim3(NB), dim3(512), 0, 0, d_rows, (const u4 *)d_idx, (const d8 *)d_w, d_out, P, nshare);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double fma = (double)NB * 8 * P * 16 * 4 * 64;
        printf("var %d rep %d: %.3f ms  %.2f TFLOP/s fp64 (vector peak 78.6)\n", var, rep, ms, 2 * fma / ms / 1e9);
    }
    std::vector<double> out((size_t)16 * 256);
    const size_t blk = 37, wv = 5, st = (blk / nshare) * 8 + wv;
    CK(hipMemcpy(out.data(), d_out + (blk * 8 + wv) * 16 * 256, out.size() * 8, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int e = 0; e < 16; e++)
        for (int n = 0; n < 256; n++) {
            double a = 0;
            for (int p = 0; p < P; p++) {
                const uint32_t word = idx[(st * P + p) * 4 + e / 4];
                const int slot = ((word >> (8 * (e % 4))) & 255) / 8;
                const double g = rows[((size_t)slot * 64 + n / 4) * 4 + (n & 3)];
                a = __builtin_fma(w[(st * P + p) * 16 + e], g, a);
            }
            const double d = fabs(a - out[e * 256 + n]);
            if (d > maxerr) maxerr = d;
        }
    printf("var %d max abs err vs host fma chain: %g\n", var, maxerr);
    }
    return 0;
}
